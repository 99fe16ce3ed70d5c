#!/usr/bin/env python3
"""Builds build/liberlamsa_hip_emu.so: the unmodified engine sources compiled with g++ against the CPU
stand-in for <hip/hip_runtime.h> in this directory.  TEST INFRASTRUCTURE (see hip/hip_runtime.h)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "build", "liberlamsa_hip_emu.so")


def build(force=False):
    src = os.path.join(ROOT, "erlamsa_amd", "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(ROOT, "include", "erlamsa_hip.h"),
                                                                 os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-x", "c++", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-attributes",
                           "-I", os.path.join(ROOT, "tests", "hipemu"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(src, "eh_engine.hip"), "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
