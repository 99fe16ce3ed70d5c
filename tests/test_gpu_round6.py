"""Round 6 GPU tests (through the C ABI, against oracle/ run LIVE on the box's host threads).

* split/1 + split_into_maxblocks/2 (erlamsa_patterns.erl:44-59): blocks of more than 1 000 000 bytes are cut at
  500 000 + rand(500 000) - 1 before the pattern walks them - inputs of 1.0 - 2.6 MB under od / nd / bu / sk.
* Cooperative execution of heavy cases (csrc/eh_common.h CoBoard): the same batches with and without it give the same bytes, and
  the posted loops are really taken by other wavefronts.
* Three contexts with batches in flight at once, and eh_result_occupancy's account of the workgroups' time.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import util  # noqa: E402
import emu_split  # noqa: E402

pytestmark = pytest.mark.gpu


def test_blocks_above_a_million_bytes_are_split_like_the_reference():
    """every input is longer than ABSMAX_BINARY_BLOCK: split_into_maxblocks/2 draws its cut points before mutate_once_loop/6 walks
    the pieces (erlamsa_patterns.erl:44-59,265-296); od, nd and bu go through split/1 on every round, sk after its skipped head"""
    import erlamsa_amd as ea
    inputs = emu_split.big_inputs(sizes=(1000001, 1000002, 1499999, 1700000, 2000001, 2600000, 3100000), per_size=3)
    checked = 0
    for pats, seed in (("od", (1, 2, 3)), ("nd,bu", (7, 7, 9)), ("sk,od", (4, 5, 6))):
        checked += emu_split.run(ea, inputs, pats, seed, engine_cap=16 << 20, big=256 << 20, oracle_threads=True)
    assert checked >= 3 * len(inputs) - 3
    print("split_into_maxblocks: %d cases bit-exact vs the live oracle" % checked)


def test_cooperative_execution_gives_the_same_bytes_and_is_used():
    """2 - 24 MiB blocks pumped by sr / lr / tr and fused (ft / fn on megabyte lists): posted copies, compares and fuse passes.
    The batch runs twice - EH_FLAG_NO_COOP and default - and once more with small chunks, so that many wavefronts take part;
    bytes, statuses and draw counts must be identical, and the board must have handed chunks to other wavefronts."""
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    rng = np.random.Generator(np.random.PCG64(606))
    inputs = []
    for k in range(96):
        per = rng.integers(97, 123, size=int(rng.integers(3, 40)), dtype=np.uint8)
        size = int(rng.integers(300000, 900000))
        b = np.tile(per, size // len(per) + 1)[:size].copy()
        for _ in range(int(rng.integers(0, 5))):
            b[int(rng.integers(0, size))] = rng.integers(0, 256)
        inputs.append(b.tobytes())
    inputs += [bytes(r) for r in synth.mixed(160, 4096, seed=66)]               # short cases between the heavy ones: the helpers
    import pyoracle as po
    data, off = po.pack(inputs)
    muts = "sr=3,lr=2,tr=1,ft=3,fn=2,bd=1,sd=1"
    runs = {}
    for name, flags, env in (("alone", ea.engine.EH_FLAG_NO_COOP, {}), ("posted", 0, {}),
                             ("small_chunks", 0, {"EH_CO_COPY_MIN": "262144", "EH_CO_COPY_CHUNK": "32768", "EH_CO_FB_MIN": "65536", "EH_CO_FB_CHUNK": "8192"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            eng = ea.Engine(0)
            eng.configure(mutations=muts, patterns="nd,bu", max_case_bytes=4 << 20, big_case_bytes=1 << 30, out_capacity=24 << 30, flags=flags)
            eng.upload_corpus(data, off)
            eng.reserve(len(inputs))                                       # (the device's pool and board exist from here on)
            before = eng.coop_stats() if name != "alone" else None
            eng.fuzz_batch(seed=(6, 0, 6))
            eng.sync()
            st, lens = eng.status().copy(), eng.lens().copy()
            dr = eng.diag()[0].copy()
            import hashlib
            sha = [hashlib.sha1(eng.fetch(i, int(lens[i]))).digest() for i in range(len(inputs))]
            after = eng.coop_stats() if name != "alone" else None
            runs[name] = (st, lens, dr, sha, None if before is None else {k: after[k] - before[k] for k in after})
            eng.close()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    a = runs["alone"]
    for name in ("posted", "small_chunks"):
        b = runs[name]
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), "%s: statuses / lengths / draw counts differ from the run without cooperation" % name
        assert a[3] == b[3], "%s: output bytes differ from the run without cooperation (cases %s)" % (name, [i for i in range(len(inputs)) if a[3][i] != b[3][i]][:8])
        d = b[4]
        assert d["loops_posted"] > 0 and d["chunks_by_helpers"] > 0, "%s: nothing was posted or no other wavefront took a chunk: %s" % (name, d)
    assert int((a[0] == 0).sum()) >= 0.9 * len(inputs)
    print("cooperative execution: %d cases identical with and without it; posted %s; small chunks %s" % (len(inputs), runs["posted"][4], runs["small_chunks"][4]))


def test_batches_of_several_contexts_in_flight_and_the_wave_slot_accounting():
    """Three contexts with batches in flight at the same time, a workgroup for every wave slot of the device each: every case comes out as the
    oracle has it, and eh_result_occupancy accounts for the workgroups - lifetimes >= time in cases + time lingering, as many workgroups as
    launched."""
    import erlamsa_amd as ea
    import pyoracle as po
    inputs = util.corpus_mixed(3000, 1024, seed=5)
    data, off = po.pack(inputs)
    muts, pats = "bd,bf,sr,lr2,tr2,ts1,num,ab,ft,len", "od,nd,bu"
    engs = []
    for k in range(3):
        e = ea.Engine(0)
        e.configure(mutations=muts, patterns=pats, max_case_bytes=4 << 20, max_slots=0)
        e.upload_corpus(data, off)
        engs.append(e)
    for k, e in enumerate(engs):
        e.fuzz_batch(seed=(3, 1, 4), first_case=1 + 3000 * k)                 # launched back to back: the three batches run side by side
    for k, e in enumerate(engs):
        want, wst, wdr, _ = po.fuzz_batch(data, off, seed=(3, 1, 4), mutations=muts, patterns=pats, first_case=1 + 3000 * k, max_case_bytes=4 << 20)
        got, gst = e.download()
        bad = [i for i in range(3000) if gst[i] == 0 and wst[i] == 0 and got[i] != want[i]]
        assert not bad and int((gst == 0).sum()) >= 2990, (k, bad[:5], int((gst == 0).sum()))
        held, in_cases, lingering, wgs, slots = e.occupancy()
        assert wgs == min(slots, 3000) and held >= in_cases + lingering and in_cases > 0, (held, in_cases, lingering, wgs, slots)
    assert engs[0].pool_stats()["contexts"] == 3
    for e in engs:
        e.close()
