#!/usr/bin/env python3
"""Adds cases to tests/golden/bench_r03.npz and tests/golden/bench_heavy_cases.json without recomputing the rest: the oracle's
(status, draws, length, SHA-1) for the given 0-based corpus rows of the bench workload (see make_bench_golden.py).
Round 4 added the cases of the last survey's 40 heaviest that the round-3 list did not hold (profiles/r04_survey_final.txt):
the cases the round's tokenizer and base64 work was about.
usage: tests/golden/extend_bench_golden.py ROW [ROW ...]"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import make_bench_golden as mk
from erlamsa_amd import synth


def main():
    rows = [int(x) for x in sys.argv[1:]]
    z = dict(np.load(os.path.join(HERE, "bench_r03.npz")))
    have = set(int(i) for i in z["idx"])
    rows = [r for r in rows if r not in have]
    if not rows:
        print("nothing to add"); return
    mat = synth.mixed(65536, 4096)
    mk.po.lib()
    res = [mk.oracle_case(mat, r) for r in rows]
    z["idx"] = np.concatenate([z["idx"], np.array(rows, dtype=np.int64)])
    z["status"] = np.concatenate([z["status"], np.array([r[0] for r in res], dtype=np.int32)])
    z["draws"] = np.concatenate([z["draws"], np.array([r[1] for r in res], dtype=np.uint64)])
    z["lens"] = np.concatenate([z["lens"], np.array([r[2] for r in res], dtype=np.uint64)])
    z["sha1"] = np.concatenate([z["sha1"], np.frombuffer(b"".join(r[3] for r in res), dtype=np.uint8).reshape(len(rows), 20)])
    np.savez_compressed(os.path.join(HERE, "bench_r03.npz"), **z)
    p = os.path.join(HERE, "bench_heavy_cases.json")
    j = json.load(open(p))
    j["cases"] += rows
    j.setdefault("added", []).append({"rows": rows, "why": "of the 40 heaviest cases of the round-4 build's survey (profiles/r04_survey_final.txt), the ones the round-3 list did not hold"})
    json.dump(j, open(p, "w"))
    print("added %d cases: %s" % (len(rows), [(r, s[0], s[2]) for r, s in zip(rows, res)]))


if __name__ == "__main__":
    main()
