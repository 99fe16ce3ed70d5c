#!/usr/bin/env python3
"""Host-side behaviour of the C ABI (call order, error codes, option parsing, buffer growth, result
accessors) exercised on the CPU emulator build.  Run by tests/test_emulated_kernel.py in a subprocess."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea
from erlamsa_amd.engine import EngineError

assert "emu" in os.environ.get("ERLAMSA_HIP_LIB", "")


def code(f, *a, **k):
    try:
        f(*a, **k)
    except EngineError as e:
        return e.code
    return 0


eng = ea.Engine(0)
# call order: nothing configured / no corpus yet
assert code(eng.fuzz_batch, seed=(1, 2, 3), n=1) == -5                       # EH_E_STATE
eng.configure(mutations="bd,bf", patterns="od")
assert code(eng.fuzz_batch, seed=(1, 2, 3), n=1) == -5                       # still no corpus
# option parsing follows erlamsa_cmdparse: unknown names, bad priorities, names the build cannot run
assert code(eng.configure, mutations="bd,nosuch") == -1                      # EH_E_INVALID
assert code(eng.configure, patterns="xx") == -1
assert code(eng.configure, mutations="sgm,js,b64") == 0                      # every mutator of the default table runs on the device
assert code(eng.configure, mutations=None, patterns=None) == 0               # NULL = the reference's default tables
assert code(eng.configure, generators="stdin=1") == -6                      # EH_E_UNSUPPORTED: host-side I/O generators
eng.configure(mutations="bd=3,bf,bi=7", patterns="od,nd=2", generators="direct=500,random=1")
inputs = util.corpus_uniform(40, 200)
data, off = po.pack(inputs)
eng.upload_corpus(data, off)
# ranges
assert code(eng.fuzz_batch, seed=(1, 2, 3), corpus_first=30, n=20) == -1     # outside the corpus
assert code(eng.fuzz_batch, seed=(1, 2, 3), first_case=0) == -1              # case numbers are 1-based
# a batch, then a smaller one, then a bigger one after reserve: buffers only grow, results stay right
want, wst, _, _ = po.fuzz_batch(data, off, seed=(1, 2, 3), mutations="bd=3,bf,bi=7", patterns="od,nd=2", generators="direct=500,random=1")
eng.fuzz_batch(seed=(1, 2, 3), n=8)
g8, s8 = eng.download()
assert g8 == want[:8] and list(s8) == list(wst[:8])
eng.reserve(40)
eng.fuzz_batch(seed=(1, 2, 3))
g, s = eng.download()
assert g == want and list(s) == list(wst)
inb, outb, nc = eng.totals()
assert nc == 40 and inb == 40 * 200 and outb == sum(map(len, want))
# the same totals summed by the kernel, and where the wave slots' time went (eh_result_summary / eh_result_occupancy: page-locked memory, no copy)
sin, sout, sn, sby = eng.summary()
assert (sin, sout, sn) == (inb, outb, nc) and int(sby[0]) == int((np.asarray(wst) == 0).sum())
held, in_cases, lingering, wgs, slots = eng.occupancy()
assert wgs >= 1 and slots >= wgs and held >= in_cases + lingering and in_cases > 0
# the download path in many chunks (device gather into two alternating bounce buffers)
eng.configure(mutations="bd=3,bf,bi=7", patterns="od,nd=2", generators="direct=500,random=1", download_chunk_bytes=700)   # results stay valid
gc, sc = eng.download()
assert gc == want and list(sc) == list(wst)
assert eng.fetch(3) == want[3] and list(eng.lens()) == [len(x) for x in want]
# a sub-range with the matching first_case reproduces the same cases
eng.fuzz_batch(seed=(1, 2, 3), first_case=11, corpus_first=10, n=5)
g5, _ = eng.download()
assert g5 == want[10:15]
# empty corpus entries and an empty batch
e2 = ea.Engine(0)
e2.configure(mutations="bd,sr,ld", patterns="od,nd,bu")
d2, o2 = po.pack([b"", b"", b"x"])
e2.upload_corpus(d2, o2)
e2.fuzz_batch(seed=(4, 5, 6))
w2, ws2, _, _ = po.fuzz_batch(d2, o2, seed=(4, 5, 6), mutations="bd,sr,ld", patterns="od,nd,bu")
g2, s2 = e2.download()
assert g2 == w2 and list(s2) == list(ws2)
e2.close(); eng.close()
# EH_FLAG_ORDERED_OUTPUT: results compacted into case order on the device; download is one contiguous
# copy and eh_result_device hands out the compact buffer with prefix-sum offsets
e3 = ea.Engine(0)
e3.configure(mutations="bd=3,bf,bi=7,sr,ld", patterns="od,nd,bu", flags=ea.engine.EH_FLAG_ORDERED_OUTPUT)
ins3 = util.corpus_mixed(150, 700, seed=3) + [b"", b"q"]
d3, o3 = po.pack(ins3)
e3.upload_corpus(d3, o3)
w3, ws3, _, _ = po.fuzz_batch(d3, o3, seed=(9, 9, 9), mutations="bd=3,bf,bi=7,sr,ld", patterns="od,nd,bu", max_case_bytes=8 << 20, max_case_work=8 << 20)
for rep in range(2):                                              # twice: the arenas swap roles every batch
    e3.fuzz_batch(seed=(9, 9, 9))
    g3, s3 = e3.download()
    assert g3 == w3 and list(s3) == list(ws3)
    dptr, optr, lptr, sptr, tot = e3.result_device()
    n3 = len(ins3)
    offs = np.ctypeslib.as_array(C.cast(optr, C.POINTER(C.c_uint64)), shape=(n3,)).copy()      # emulator: device memory is host memory
    lens = np.ctypeslib.as_array(C.cast(lptr, C.POINTER(C.c_uint64)), shape=(n3,)).copy()
    assert tot == sum(map(len, w3)) and list(lens) == [len(x) for x in w3]
    assert list(offs) == list(np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64))
    flat = bytes(np.ctypeslib.as_array(C.cast(dptr, C.POINTER(C.c_uint8)), shape=(max(tot, 1),))[:tot])
    assert flat == b"".join(w3)
e3.close()
# ---- erlamsa_out's file sink (erlamsa_out.erl:103-123): "%n" -> case number, one file per EH_CASE_OK case, bytes = the case's output
import tempfile
with tempfile.TemporaryDirectory() as td:
    ef = ea.Engine(0)
    ef.configure(mutations="bd,bf,sr,num,ld", patterns="od,nd,bu")
    ef.upload_corpus(data, off)
    ef.fuzz_batch(seed=(4, 4, 4), first_case=11)
    gotf, stf = ef.download()
    nf, nb, ns = ef.write_files(os.path.join(td, "case-%n-of-%n.bin"), first_number=11, threads=3)
    okc = [i for i in range(len(gotf)) if stf[i] == 0]
    assert nf == len(okc) and ns == len(gotf) - len(okc) and nb == sum(len(gotf[i]) for i in okc)
    for i in okc:
        with open(os.path.join(td, "case-%d-of-%d.bin" % (11 + i, 11 + i)), "rb") as fh:
            assert fh.read() == gotf[i], i
    assert len(os.listdir(td)) == nf
    assert code(ef.write_files, os.path.join(td, "no-such-dir", "x-%n")) == -1          # "Error opening file ..."
    ef.close()
# introspection tables mirror the reference's tables
names = [m[0] for m in ea.mutator_table()]
assert names[0] == "sgm" and names[-1] == "nil" and len(names) == 41
assert [p[0] for p in ea.pattern_table()] == ["od", "nd", "bu", "sk", "sz", "cs", "ar", "cp", "co", "nu"]
# ---- request coalescing: tickets from interleaved submits get what each request gets alone (eh_fuzz_calls semantics)
ec = ea.Engine(0)
ec.configure(mutations="bd,bf,sr,num,ld", patterns="od,nd,bu")
reqs = [(inputs[i % len(inputs)], (100 + i, 7 * i + 1, 3 * i + 2)) for i in range(23)]
seeds = np.array([s for _, s in reqs], dtype=np.int64)
dq, oq = po.pack([b for b, _ in reqs])
wantq, wstq, _, _ = po.fuzz_batch(dq, oq, seeds=seeds, mutations="bd,bf,sr,num,ld", patterns="od,nd,bu")
ec.coalesce_limits(8, 1 << 20)                       # a launch every 8 requests
tickets = [ec.submit(b, s) for b, s in reqs]          # 23 requests: two full batches launched, 7 pending
assert len(set(tickets)) == 23
assert ec.poll(tickets[-1]) is None                   # still pending: EH_E_AGAIN
# a context with coalesced requests pending or in flight belongs to the coalescer: batches, corpora, configurations are refused
assert code(ec.fuzz_batch, seed=(1, 2, 3)) == -5 and code(ec.upload_corpus, dq, oq) == -5 and code(ec.configure, mutations="bd") == -5
assert ec.poll(tickets[3]) == (int(wstq[3]), wantq[3])      # first batch (already collected when the second was launched)
assert ec.poll(tickets[12]) == (int(wstq[12]), wantq[12])   # second batch: waits for it
ec.flush()
for k in (22, 0, 17, 9):
    assert ec.poll(tickets[k]) == (int(wstq[k]), wantq[k]), k
assert code(ec.poll, tickets[0]) == -1                # consumed
assert code(ec.poll, 999999) == -1                    # unknown
ec.flush()                                            # nothing pending: no-op
# eh_cancel: a pending request leaves its batch, a launched one is dropped at collection, a finished one is freed
tk = [ec.submit(b, s) for b, s in reqs[:5]]
ec.cancel(tk[1])                                      # pending
ec.flush()
ec.cancel(tk[3])                                      # launched
assert ec.poll(tk[0]) == (int(wstq[0]), wantq[0]) and ec.poll(tk[2]) == (int(wstq[2]), wantq[2])
assert code(ec.poll, tk[1]) == -1 and code(ec.poll, tk[3]) == -1
ec.cancel(tk[4])                                      # finished, never polled
assert code(ec.poll, tk[4]) == -1 and code(ec.cancel, tk[4]) == -1
ec.fuzz_batch(seed=(1, 2, 3))                         # the coalescer is idle again: the context takes batches
ec.close()
# a request with OTHER options than the previous one (what erlang/c_src nif_submit does: eh_flush, eh_configure, eh_submit): the
# pending request is launched with the options it came with, eh_configure collects that batch instead of refusing, both poll fine
eo = ea.Engine(0)
eo.configure(mutations="bd,bf,sr,num,ld", patterns="od,nd,bu")
ta = eo.submit(reqs[0][0], reqs[0][1])
assert code(eo.configure, mutations="bi,br", patterns="od") == -5      # still pending: refused
eo.flush()                                             # launched: in flight, nobody has polled
eo.configure(mutations="bi,br", patterns="od")         # collects the batch in flight, then the options change
tb = eo.submit(reqs[1][0], reqs[1][1])
eo.flush()
d1, o1 = po.pack([reqs[1][0]])
wb, wsb, _, _ = po.fuzz_batch(d1, o1, seeds=np.array([reqs[1][1]], dtype=np.int64), mutations="bi,br", patterns="od")
assert eo.poll(tb) == (int(wsb[0]), wb[0])
assert eo.poll(ta) == (int(wstq[0]), wantq[0])         # the first request kept the options it was submitted under
eo.close()
# the option-map API: skip => N drops the cases numbered <= N, whose draws still happen (erlamsa_main.erl:161,191-196), also when
# the run does not start at case 1; keys that need the BEAM are refused
from erlamsa_amd import api
base = {"seed": (4, 5, 6), "mutations": "bd,bf,bi,sr", "patterns": "od,nd", "input": b"skip me, the quick brown fox", "n": 12}
all12 = api.fuzzer(dict(base, on_engine_limit="skip"))
outs12, st12 = api.fuzz_batch([base["input"]] * 12, base, return_status=True)
kept = [o for o, s_ in zip(outs12, st12) if s_ == 0 and len(o) > 0]
assert all12 == kept
dropped5 = [o for i, (o, s_) in enumerate(zip(outs12, st12)) if i >= 5 and s_ == 0 and len(o) > 0]
assert api.fuzzer(dict(base, skip=5, on_engine_limit="skip")) == dropped5
late = api.fuzz_batch([base["input"]] * 6, dict(base, first_case=4), return_status=True)
assert api.fuzzer(dict(base, n=6, first_case=4, skip=5, on_engine_limit="skip")) == [o for i, (o, s_) in enumerate(zip(*late)) if 4 + i > 5 and s_ == 0 and len(o) > 0]
try:
    api.fuzzer(dict(base, external_post="external_post"))
    raise SystemExit("external_post was not refused")
except api.Unsupported as e:
    assert e.keys == ["external_post"]
print("abi behaviour ok")
