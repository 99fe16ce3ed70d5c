"""Golden fixtures (tests/golden/vectors.json, made by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them.  GPU: the engine reproduces them through the C ABI without
the oracle being present."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _vectors():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)["vectors"]


@pytest.mark.parametrize("vec", _vectors(), ids=lambda v: v["name"])
def test_oracle_reproduces_golden(vec):
    import pyoracle as po
    inputs = [bytes.fromhex(h) for h in vec["inputs_hex"]]
    data, off = po.pack(inputs)
    outs, st, _, _ = po.fuzz_batch(data, off, seed=tuple(vec["seed"]), mutations=vec["mutations"], patterns=vec["patterns"],
                                   first_case=vec["first_case"], max_case_bytes=256 << 10)
    assert [o.hex() for o in outs] == vec["outputs_hex"]
    assert [int(x) for x in st] == vec["status"]


@pytest.mark.gpu
@pytest.mark.parametrize("vec", _vectors(), ids=lambda v: v["name"])
def test_engine_reproduces_golden(vec):
    import erlamsa_amd as ea
    inputs = [bytes.fromhex(h) for h in vec["inputs_hex"]]
    outs, st = ea.fuzz_batch(inputs, {"seed": tuple(vec["seed"]), "mutations": vec["mutations"], "patterns": vec["patterns"],
                                      "first_case": vec["first_case"]}, return_status=True)
    for i, (o, s) in enumerate(zip(outs, st)):
        if vec["status"][i] in (2, 3) or s in (2, 3):
            continue          # work-area cap / container re-encode paths
        assert o.hex() == vec["outputs_hex"][i], "case %d of %s" % (i, vec["name"])
        assert int(s) == vec["status"][i]
