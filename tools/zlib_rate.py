#!/usr/bin/env python3
"""Throughput of the device codecs (csrc/eh_zlib.h) as the container patterns use them: N gzip inputs of S plain bytes each through
pattern cp (decode, mutate, re-encode; one wavefront per case, the codecs on one lane) - cases/s and plain MB/s - and one stream
alone through eh_selftest_zlib (single-lane latency).  One JSON line.  NOT run in round 3: no GPU minutes were left.

  python tools/zlib_rate.py [--cases 4096] [--size 4096]
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=4096)
    ap.add_argument("--size", type=int, default=4096)
    a = ap.parse_args()
    plain = [bytes(r) for r in synth.mixed(a.cases, a.size, seed=5)]
    gz = []
    for p in plain:
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 8)
        gz.append(c.compress(p) + c.flush())
    data, off = ea.pack_corpus(gz)
    eng = ea.Engine(0)
    eng.configure(mutations="bd,bf,bi,sr,num,lr", patterns="cp", max_case_bytes=4 << 20)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(1, 2, 3)); eng.sync()                    # warm-up (allocations)
    t0 = time.perf_counter()
    eng.fuzz_batch(seed=(4, 5, 6)); eng.sync()
    dt = time.perf_counter() - t0
    st = np.bincount(eng.status(), minlength=6).tolist()
    res = {"cp_pattern": {"cases": a.cases, "plain_bytes_per_case": a.size, "seconds": round(dt, 4), "cases_per_s": round(a.cases / dt, 1),
                          "plain_MB_per_s_decoded_plus_encoded": round(2 * a.cases * a.size / dt / 1e6, 2), "kernel_ms": eng.kernel_ms(), "status": st}}
    one = {}
    for name, blob in (("text", plain[1] if len(plain) > 1 else plain[0]), ("random", np.random.default_rng(1).integers(0, 256, a.size, dtype=np.uint8).tobytes())):
        t0 = time.perf_counter(); c = eng.selftest_zlib(1, blob); t1 = time.perf_counter(); d = eng.selftest_zlib(4, c); t2 = time.perf_counter()
        assert d == blob
        one[name] = {"bytes": len(blob), "deflate_ms": round(1e3 * (t1 - t0), 3), "inflate_ms": round(1e3 * (t2 - t1), 3)}
    res["single_stream_one_lane_incl_launch_and_copies"] = one
    eng.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
