"""The multi-GPU entry points of the C ABI (ABI 7: eh_comm_*, eh_corpus_broadcast, eh_corpus_allgather, eh_corpus_broadcast_local)
on CPU ranks: the emulator build of the engine ("device memory" is host memory) with tests/hipemu/fake_rccl.cpp standing in for
librccl.so (EH_RCCL_LIB).  What is checked is the library's own logic around the collectives - sizes exchanged, buffers owned,
offsets rebuilt, shards in rank order - and that case-range sharding over W ranks gives the bytes of a 1-rank run (strong scaling:
ONE run of n cases split by shard.case_range, erlamsa_main.erl:95-108).  RCCL itself runs in the -m gpu tests
(tests/test_gpu_round5_comm.py)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
FAKE = os.path.join(ROOT, "build", "libfake_rccl.so")


def _build():
    import build_emu
    emu = build_emu.build()
    src = os.path.join(ROOT, "tests", "hipemu", "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", src, "-o", FAKE, "-lrt", "-pthread"])
    return emu


RANK_SCRIPT = r'''
import hashlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import shard, synth
rank, world, mode, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
n, size = 48, 600
mat = synth.mixed(n, size, seed=11)
e = ea.Engine(0)
e.configure(mutations="bd,bf,bi,sr,sd,num,ld,lr,ft,fn,fo,len", patterns="od,nd,bu,sz", generators="direct=5,jump=3" if mode == "allgather" else None, max_case_bytes=1 << 20)
single = mode.startswith("single")                    # the 1-rank reference run: no communicator, eh_corpus_upload
if single:
    e.upload_corpus(*synth.as_arena(mat))
elif rank == 0:
    open(idfile + ".tmp", "wb").write(ea.Engine.comm_unique_id()); os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        time.sleep(0.01)
        assert time.time() - t0 < 60
if not single:
    e.comm_init(open(idfile, "rb").read(), rank, world)
if single:
    pass
elif mode == "broadcast":
    root = world - 1                                   # not rank 0: the root is an argument, not a convention
    if rank == root:
        e.corpus_broadcast(root, *synth.as_arena(mat))
    else:
        e.corpus_broadcast(root)
else:                                                  # every rank brings its shard of n / world rows
    per = n // world
    e.corpus_allgather(*synth.as_arena(mat[rank * per:(rank + 1) * per]))
d, o, cn, nb = e.corpus_device()
import ctypes
arena = ctypes.string_at(d, nb); offs = np.frombuffer(ctypes.string_at(o, 8 * (cn + 1)), dtype=np.uint64)
first, cnt = shard.case_range(n, rank, world)
e.fuzz_batch(seed=(9, 8, 7), first_case=first + 1, corpus_first=first, n=cnt)
outs, st = e.download()
json.dump({"arena_sha1": hashlib.sha1(arena).hexdigest(), "offs_ok": bool((offs == np.arange(cn + 1, dtype=np.uint64) * size).all()), "n": int(cn),
           "first": first, "sha1": [hashlib.sha1(x).hexdigest() for x in outs], "status": [int(x) for x in st]}, open(out, "w"))
if not single:
    e.comm_destroy()
e.close()
'''


@pytest.mark.parametrize("mode,world", [("broadcast", 3), ("allgather", 2)])
def test_ranks_load_the_arena_through_the_library_and_shard_the_run(tmp_path, mode, world):
    emu = _build()
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu, EH_RCCL_LIB=FAKE)
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT % {"root": ROOT})
    idfile = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), mode, idfile, str(tmp_path / ("r%d.json" % r))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(world)]
    # one rank, no collectives: the reference run
    ref = subprocess.run([sys.executable, str(script), "0", "1", "single_" + mode, idfile, str(tmp_path / "single.json")], env=env, capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stdout[-1500:] + ref.stderr[-1500:]
    one = json.load(open(tmp_path / "single.json"))
    want, st = one["sha1"], one["status"]
    from erlamsa_amd import synth
    data, _ = synth.as_arena(synth.mixed(48, 600, seed=11))
    for r in res:
        assert r["n"] == 48 and r["offs_ok"] and r["arena_sha1"] == hashlib.sha1(data.tobytes()).hexdigest(), "rank holds another arena"
        assert r["sha1"] == want[r["first"]:r["first"] + len(r["sha1"])], "a rank's cases differ from the 1-rank run"
        assert r["status"] == st[r["first"]:r["first"] + len(r["sha1"])]
    assert sum(len(r["sha1"]) for r in res) == 48


def test_one_process_several_devices_broadcast_local_and_refusals(tmp_path):
    emu = _build()
    code = r'''
import sys, ctypes, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import shard, synth
mat = synth.mixed(64, 500, seed=5); data, off = synth.as_arena(mat)
es = [ea.Engine(d) for d in range(4)]
for e in es: e.configure(mutations="bd,bf,sr,num,lr,ft", patterns="od,nd")
try:
    ea.Engine.corpus_broadcast_local(es, 0); raise SystemExit("broadcast_local without communicators must be refused")
except ea.EngineError as ex: assert ex.code == -5, ex
ea.Engine.comm_init_local(es)
es[2].upload_corpus(data, off)
ea.Engine.corpus_broadcast_local(es, 2)
got = []
for r, e in enumerate(es):
    d, o, cn, nb = e.corpus_device()
    assert cn == 64 and ctypes.string_at(d, nb) == data.tobytes()
    first, cnt = shard.case_range(64, r, 4)
    e.fuzz_batch(seed=(3, 1, 4), first_case=first + 1, corpus_first=first, n=cnt)
for e in es: got += [hashlib.sha1(x).hexdigest() for x in e.download()[0]]
one = ea.Engine(0); one.configure(mutations="bd,bf,sr,num,lr,ft", patterns="od,nd"); one.upload_corpus(data, off); one.fuzz_batch(seed=(3, 1, 4))
assert got == [hashlib.sha1(x).hexdigest() for x in one.download()[0]]
# two contexts of ONE device share a corpus with eh_corpus_device / eh_corpus_attach, not with a communicator
try:
    ea.Engine.comm_init_local([ea.Engine(1), ea.Engine(1)]); raise SystemExit("two contexts of one device must be refused")
except ea.EngineError as ex: assert ex.code == -1, ex
# sequence_muta chains the scores over the cases of a run: refused, not silently ignored
try:
    one.configure(sequence_muta=True); raise SystemExit("sequence_muta must be refused")
except ea.EngineError as ex: assert ex.code == -6 and "sequence_muta" in str(ex), ex
print("ok")
''' % {"root": ROOT}
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu, EH_RCCL_LIB=FAKE, HIPEMU_DEVICES="4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_a_host_without_rccl_gets_an_error_code_not_a_crash():
    """include/erlamsa_hip.h: EH_E_UNSUPPORTED when librccl cannot be loaded (a single-GPU host has no need of it) - from every entry
    point that would call it, with the loader's text where there is a context to keep it."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    code = r'''
import sys
sys.path.insert(0, %r)
import erlamsa_amd as ea
seen = []
try:
    ea.Engine.comm_unique_id()
except ea.EngineError as e:
    seen.append(e.code)
e = ea.Engine(0)
try:
    e.comm_init(bytes(128), 0, 1)
except ea.EngineError as x:
    seen.append(x.code); assert "cannot load RCCL" in str(x) and "no_such_rccl" in str(x), str(x)
try:
    ea.Engine.comm_init_local([e])
except ea.EngineError as x:
    seen.append(x.code)
print("codes", seen)
''' % ROOT
    env = dict(os.environ, ERLAMSA_HIP_LIB=build_emu.build(), EH_RCCL_LIB=os.path.join(ROOT, "build", "no_such_rccl.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "codes [-6, -6, -6]" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
