#!/usr/bin/env python3
"""Worst-case private stack per lane of eh_mutate_kernel, from the assembler's own numbers.

The kernel recurses (nested scheduler calls of b64 / sgm / js: muta_X -> nested_fuzz -> muta_X ..., at most MAX_NEST nested_fuzz
frames), so the compiler cannot bound its stack and the runtime sizes the scratch from hipLimitStackSize (eh_create: 6 144 bytes).
This script compiles csrc/eh_engine.hip to assembly, reads every function's `.private_seg_size` expression (own frame + max over
callees; the compiler leaves the recursive edge out), and evaluates the deepest chain:

    kernel's own frame + MAX_NEST x (heaviest mutator that nests + nested_fuzz) + heaviest leaf mutator with all it calls

usage: tools/stack_chain.py [file.s]      (without a file: compiles first, ~2 min)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-Xclang", "-target-feature", "-Xclang", "-mai-insts", "-fPIC", "-S", "--cuda-device-only"]
MAX_NEST, LIMIT = 6, 6144


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", path, os.path.join(ROOT, "erlamsa_amd", "csrc", "eh_engine.hip")], check=True, stderr=subprocess.DEVNULL)
    own, calls = {}, {}
    for m in re.finditer(r"\.set\s+(\S+?)\.private_seg_size,\s*(.*)", open(path).read()):
        name, expr = m.group(1), m.group(2)
        short = re.sub(r"^\.L", "", name); short = re.sub(r"^_ZN2eh\d+", "", short); short = re.sub(r"E[RPNiIjmb].*$", "", short)
        mm = re.match(r"(\d+)(?:\+max\((.*)\))?", expr)
        own[short] = int(mm.group(1))
        calls[short] = [re.sub(r"E[RPNiIjmb].*$", "", re.sub(r"^\.?L?_?Z?N?2?e?h?\d*", "", re.sub(r"\.private_seg_size", "", c.strip()))) for c in (mm.group(2) or "").split(",") if c.strip() and not c.strip().isdigit()]
        calls[short] += []                                   # literal numbers inside max() are frames of callees already folded in
        lit = [int(c) for c in (mm.group(2) or "").split(",") if c.strip().isdigit()]
        own[short + "#lit"] = max(lit) if lit else 0

    def total(f, stop=("nested_fuzz",), seen=()):
        """own frame + deepest callee, not following the recursive edge"""
        if f not in own or f in seen:
            return 0
        return own[f] + max([own.get(f + "#lit", 0)] + [total(c, stop, seen + (f,)) for c in calls.get(f, []) if c not in stop])

    kernel = next(k for k in own if k.startswith("eh_mutate_kernel"))
    nesters = {"muta_sgml": own["muta_sgml"], "muta_json": own["muta_json"], "muta_b64": own["muta_b64"]}
    leaves = {f: total(f) for f in own if f.startswith("muta_")}
    level = max(nesters.values()) + own["nested_fuzz"]
    leaf = max(leaves.values())
    chain = own[kernel] + MAX_NEST * level + leaf
    print("own frames: kernel %d, nested_fuzz %d, nesting mutators %s" % (own[kernel], own["nested_fuzz"], nesters))
    print("leaf mutators with all they call: %s" % dict(sorted(leaves.items(), key=lambda kv: -kv[1])[:6]))
    print("deepest chain: %d + %d x (%d + %d) + %d = %d bytes per lane; hipLimitStackSize %d (%d to spare)" % (
        own[kernel], MAX_NEST, max(nesters.values()), own["nested_fuzz"], leaf, chain, LIMIT, LIMIT - chain))
    sys.exit(0 if chain <= LIMIT else 1)


if __name__ == "__main__":
    main()
