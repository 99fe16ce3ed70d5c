#!/usr/bin/env python3
"""Do two builds of liberlamsa_hip.so hold the same gfx950 instruction stream?  Compares the .text sections of the device code objects:
the disassembly with every literal replaced by a placeholder, then the raw words (what differs there are pc-relative literals; their
deltas are printed).  Used in round 4 to show that making the constant tables compile-time data changed no instruction of
eh_mutate_kernel: the profiles of the build before it are profiles of the build after it.
usage: tools/same_kernel_text.py old.so new.so"""
import collections
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def device_elf(path):
    d = open(path, "rb").read()
    b = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
    e = [m.start() for m in re.finditer(b"\x7fELF", d) if m.start() > b][0]
    f = tempfile.NamedTemporaryFile(suffix=".elf", delete=False)
    f.write(d[e:]); f.close()
    return f.name


def text_of(elf):
    o = subprocess.run([LLVM + "llvm-readelf", "-S", elf], capture_output=True, text=True).stdout
    m = re.search(r"\]\s+\.text\s+\w+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", o)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    return open(elf, "rb").read()[off:off + size]


def disasm(elf):
    o = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", elf], capture_output=True, text=True).stdout
    out = []
    for ln in o.splitlines()[2:]:
        ln = re.sub(r"0x[0-9a-fA-F]+", "IMM", ln); ln = re.sub(r"//.*$", "", ln); ln = re.sub(r"<[^>]*>", "", ln)
        out.append(ln.rstrip())
    return out


a, b = device_elf(sys.argv[1]), device_elf(sys.argv[2])
da, db = disasm(a), disasm(b)
same = da == db
print("instructions: %d vs %d lines, %s" % (len(da), len(db), "identical up to literals" if same else "DIFFERENT"))
ta, tb = text_of(a), text_of(b)
if len(ta) == len(tb):
    w = [i for i in range(0, len(ta), 4) if ta[i:i + 4] != tb[i:i + 4]]
    d = collections.Counter(struct.unpack_from("<i", tb, i)[0] - struct.unpack_from("<i", ta, i)[0] for i in w)
    print(".text %d bytes, %d words differ; deltas: %s" % (len(ta), len(w), dict(d.most_common(6))))
else:
    print(".text sizes differ: %d vs %d" % (len(ta), len(tb)))
sys.exit(0 if same else 1)
