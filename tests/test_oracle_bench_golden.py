"""tests/golden/bench_r03.npz (what test_gpu_parity.py::test_bench_workload_full_table_vs_oracle compares the engine with) is
re-derived from the live oracle on a sample: the file cannot drift from oracle/ unnoticed."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bench_golden_matches_the_live_oracle_on_a_sample():
    import pyoracle as po
    from erlamsa_amd import synth
    z = np.load(os.path.join(HERE, "golden", "bench_r03.npz"))
    assert "oracle" in str(z["generator"])
    heavy = json.load(open(os.path.join(HERE, "golden", "bench_heavy_cases.json")))["cases"]
    assert set(int(i) for i in z["idx"]) >= set(range(4096)) | set(heavy)
    pos = {int(i): k for k, i in enumerate(z["idx"])}
    mat = synth.mixed(65536, 4096)
    # 64 consecutive rows in one call, plus two single heavy cases with outputs below 8 MB
    d, o = synth.as_arena(mat[1000:1064])
    outs, st, dr, _ = po.fuzz_batch(d, o, seed=(1, 2, 3), patterns="od,nd,bu", first_case=1001, max_case_bytes=1 << 30)
    for j in range(64):
        k = pos[1000 + j]
        assert int(st[j]) == int(z["status"][k]) and int(dr[j]) == int(z["draws"][k]) and len(outs[j]) == int(z["lens"][k])
        assert hashlib.sha1(outs[j]).digest() == z["sha1"][k].tobytes()
    small = [i for i in heavy if int(z["lens"][pos[i]]) < (8 << 20)][:2]
    for i in small:
        d, o = synth.as_arena(mat[i:i + 1])
        outs, st, dr, _ = po.fuzz_batch(d, o, seed=(1, 2, 3), patterns="od,nd,bu", first_case=i + 1, max_case_bytes=1 << 30)
        k = pos[i]
        assert int(st[0]) == int(z["status"][k]) and int(dr[0]) == int(z["draws"][k]) and hashlib.sha1(outs[0]).digest() == z["sha1"][k].tobytes()
