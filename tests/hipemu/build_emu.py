#!/usr/bin/env python3
"""Builds build/liberlamsa_hip_emu.so: the unmodified engine sources compiled with g++ against the CPU
stand-in for <hip/hip_runtime.h> in this directory.  TEST INFRASTRUCTURE (see hip/hip_runtime.h)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "build", "liberlamsa_hip_emu.so")


OUT_RACE = os.path.join(ROOT, "build", "liberlamsa_hip_emu_race.so")


def build(force=False, race=False):
    """race=True: the same sources with every load / store of the kernel code calling into race_hooks.cpp (see there)"""
    src = os.path.join(ROOT, "erlamsa_amd", "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(ROOT, "include", "erlamsa_hip.h"),
                                                                 os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h")]
    out = OUT_RACE if race else OUT
    hooks = os.path.join(ROOT, "tests", "hipemu", "race_hooks.cpp")
    if race:
        deps.append(hooks)
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    base = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-attributes",
            "-I", os.path.join(ROOT, "tests", "hipemu"), "-I", os.path.join(ROOT, "include")]
    if not race:
        subprocess.check_call(base + ["-x", "c++", "-shared", os.path.join(src, "eh_engine.hip"), "-o", out])
        return out
    obj = [out + ".engine.o", out + ".hooks.o"]
    subprocess.check_call(base + ["-DHIPEMU_RACE=1", "-fsanitize=kernel-address", "--param", "asan-instrumentation-with-call-threshold=0",
                                  "--param", "asan-globals=0", "--param", "asan-stack=0", "-x", "c++", "-c", os.path.join(src, "eh_engine.hip"), "-o", obj[0]])
    subprocess.check_call(base + ["-c", hooks, "-o", obj[1]])
    subprocess.check_call(["g++", "-shared", "-rdynamic"] + obj + ["-o", out])
    for o in obj:                                   # (11 MB of objects would travel with every gpurun snapshot)
        os.remove(o)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, race="--race" in sys.argv))
