#!/usr/bin/env python3
"""Rewrite tests/golden/vectors.json from a BEAM capture (the output of capture.escript): outputs_hex become what the real
erlamsa produced and `generator` says so.  BEAM cannot tell an empty result from a dead worker (both are <<>>, see the
escript), so `status` keeps the oracle's 0/1 where the output is empty and becomes 0 elsewhere; cases the oracle itself
could not run (status 2/3) keep their marker and are skipped by the tests.

usage: apply_beam_capture.py beam_capture.txt"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lines = [ln.split() for ln in open(sys.argv[1]) if ln.strip()]
    assert lines and lines[0][0] == "otp", "not a capture.escript output"
    otp = lines[0][1]
    cap = {}
    for ln in lines[1:]:
        cap[(ln[0], int(ln[1]))] = ln[2] if len(ln) > 2 else ""
    path = os.path.join(HERE, "vectors.json")
    with open(path) as f:
        doc = json.load(f)
    changed = 0
    for v in doc["vectors"]:
        if "generators" in v:                            # file / jump sets are not part of the capture (see make_golden.py)
            continue
        for i in range(len(v["inputs_hex"])):
            if v["status"][i] in (2, 3):
                continue
            new = cap[(v["name"], i)]
            if new != v["outputs_hex"][i]:
                changed += 1
            v["outputs_hex"][i] = new
            if new:
                v["status"][i] = 0
    doc["generator"] = "BEAM: erlamsa_main:fuzzer/1 via tests/golden/capture.escript (OTP %s)" % otp
    with open(path, "w") as f:
        json.dump(doc, f, indent=0)
    print("vectors.json now holds BEAM outputs (OTP %s); %d case(s) differed from the previous file" % (otp, changed))


if __name__ == "__main__":
    main()
