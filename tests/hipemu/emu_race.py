#!/usr/bin/env python3
"""The engine under the emulator's RACE DETECTOR (build_emu.py --race, tests/hipemu/race_hooks.cpp): every load and store of the
kernel code is checked against the rule that, between two rendezvous points of the wavefront, no lane reads a byte another lane
wrote or overwrites a byte another lane read.  The plain emulator cannot see such a bug (lane 0 runs first, so the others do see
what it wrote); a real wavefront promises nothing without the wave_sync().  Workloads: the file / jump / random generators, gzip /
zlib / zip containers (their codecs run on lane 0 and hand results to the wavefront), nearly full and tiny slots with the default
tables (nested scheduler calls, areas borrowed and returned), the streaming fuse with the meta trace, the sgm tokenizer's lane
batches and the base64 decode by the wave (round 4).  The same hooks check
BOUNDS: in this build every device allocation has a guard zone on either side, and an access that lands in one is reported.

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu_race.so python tests/hipemu/emu_race.py [cases per workload]
"""
import ctypes
import os
import subprocess
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea
from erlamsa_amd import synth
import emu_containers as ec

warnings.simplefilter("ignore")
assert "race" in os.environ.get("ERLAMSA_HIP_LIB", ""), "point ERLAMSA_HIP_LIB at the race build (build_emu.py --race)"
lib = ctypes.CDLL(os.environ["ERLAMSA_HIP_LIB"])
lib.hipemu_race_count.restype = ctypes.c_ulong
lib.hipemu_oob_count.restype = ctypes.c_ulong

# the bounds check is alive: a byte-mover job that reads 104 bytes past its 4 096-byte device buffer lands in the guard zone
_e = ea.Engine(0)
_e.selftest_movers(np.zeros(4096, dtype=np.uint8), np.array([[0, 0, 4000, 200, 0]], dtype=np.uint32))
_e.close()
OOB0 = lib.hipemu_oob_count()
assert OOB0 > 0, "the seeded overrun was not reported"


def run(tag, inputs, generators=None, slot=1 << 20, flags=0, fsm=0, **kw):
    data, off = po.pack(inputs)
    eng = ea.Engine(0)
    eng.configure(generators=generators, max_case_bytes=slot, big_case_bytes=32 << 20, flags=flags, fuse_stream_min=fsm, **kw)
    eng.upload_corpus(data, off); eng.fuzz_batch(seed=(3, 1, 4)); _, st = eng.download()
    if flags:
        for i in range(len(inputs)):
            eng.meta(i)
    eng.close()
    print("%-28s cases %3d statuses %s races so far %d" % (tag, len(inputs), np.bincount(st, minlength=4).tolist(), lib.hipemu_race_count()), flush=True)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
run("generator file", util.corpus_mixed(n, 9000, seed=1), generators="file", mutations="bd,sr,num,lr,ft", patterns="od,nd,bu,sz,cs,sk")
run("generator jump", util.corpus_mixed(n, 9000, seed=2), generators="jump", mutations="bd,sr,num,lr,fo", patterns="od,nd,bu,sz")
run("random generator, full slot", util.corpus_uniform(n, 64, seed=3), generators="random=1", mutations="num,bd,lr", patterns="cs,sz,sk,od,nd", slot=32768)
run("gzip / zlib through cp", ec.compressed_corpus(2 * n, 300), mutations="bd,bf,sr,num,lr", patterns="cp,sz,od")
run("zip through ar and zip", ec.zip_corpus(2 * n, 500), mutations="zip=3,bd,sr,num", patterns="ar=3,od,nd")
run("default tables, 16 KiB slots", util.corpus_mixed(n, 700, seed=9) + synth.sgml_docs(2, seed=5) + synth.json_docs(2, seed=6), slot=16384)
run("streaming fuse + meta trace", util.corpus_mixed(n, 3000, seed=4), mutations="ft,fn,fo,bd", patterns="od,nd,bu", fsm=64, flags=ea.engine.EH_FLAG_META_TRACE)
import emu_sgml_replay
import emu_b64
# round 4: tag attempts one per lane (pieces, tokens and memo marks written by the lanes, read by the wave-wide machine), base64 chunks
# packed and decoded by the wave
run("sgm: lane batches", emu_sgml_replay.corpus(1, 21, 1, small=True)[:3], mutations="sgm", patterns="od", slot=8 << 20)
run("b64: decode by the wave", emu_b64.corpus(1, 21)[1:3], mutations="b64", patterns="od", slot=8 << 20)
races = lib.hipemu_race_count()
if races:
    pcs = (ctypes.c_void_p * 256)(); cnt = (ctypes.c_ulong * 256)()
    k = lib.hipemu_race_sites(pcs, cnt, 256)
    base = int([ln for ln in open("/proc/self/maps").read().splitlines() if os.path.basename(os.environ["ERLAMSA_HIP_LIB"]) in ln][0].split("-")[0], 16)
    for i in range(k):
        o = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", os.environ["ERLAMSA_HIP_LIB"], "0x%x" % (pcs[i] - base - 1)], capture_output=True, text=True).stdout.split("\n")
        print("%8d  %s" % (cnt[i], " <- ".join("%s@%s" % (o[j].split("(")[0], os.path.basename(o[j + 1]).split(" ")[0]) for j in range(0, min(len(o) - 1, 8), 2))))
    sys.exit("%d cross-lane accesses without a rendezvous in between" % races)
oob = lib.hipemu_oob_count() - OOB0
if oob:
    sys.exit("%d accesses outside a device allocation (guard zones)" % oob)
print("race ok")
