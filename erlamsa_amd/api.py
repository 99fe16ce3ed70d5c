"""Option-map level API mirroring erlamsa_main:fuzzer/1 and erlamsa_app:fuzz/1,2.

Option keys are the reference's Dict keys (erlamsa_main.erl:127-163):
  seed        {A,B,C} tuple                      (default: os.urandom, like gen_urandom_seed/0)
  mutations   [(name, pri)] or "-m" string       (default erlamsa_mutations:default/1)
  patterns    [(name, pri)] or "-p" string       (default erlamsa_patterns:default/0)
  generators  [(name, pri)]                      (default: what paths=[direct] leaves: direct=500, random=1)
  n           number of cases                    (default 1)
  input       bytes                              (paths = [direct])
  blockscale  float
  skip        cases numbered <= skip are computed but dropped (erlamsa_main.erl:161,191-196)
Only `paths => [direct]`, `output => return` is served by the GPU path.  Keys the reference honours and a batch
on the GPU cannot - external_mutations (custom mutator funs appended to the table, erlamsa_main.erl:128), external_post (a
post-processor applied per written block, :159), sequence_muta (:223-235) - raise Unsupported: the caller routes that run to
the BEAM path instead of getting a run WITHOUT what it asked for (erlang/src/erlamsa_hip.erl host_only/1 is the same rule).
"""
import os

import numpy as np

from .engine import CASE_OK, Engine, mutator_table, pattern_table

_engines = {}


def _engine(device=0):
    e = _engines.get(device)
    if e is None:
        e = _engines[device] = Engine(device)
    return e


def default_mutations():
    """erlamsa_mutations:default/1 -> [(name, pri)] (erlamsa_mutations.erl:1358-1359)"""
    return [(n, p) for n, p, _ in mutator_table()]


def default_patterns():
    """erlamsa_patterns:default/0 (erlamsa_patterns.erl:407-408)"""
    return [(n, p) for n, p, _ in pattern_table()]


def actions_to_string(lst):
    if lst is None or isinstance(lst, str):
        return lst
    return ",".join("%s=%d" % (n, p) for n, p in lst)


def pack_corpus(inputs):
    """list[bytes] -> (uint8 arena, uint64 off[n+1])  — the packed offset/length arena"""
    off = np.zeros(len(inputs) + 1, dtype=np.uint64)
    if inputs:
        off[1:] = np.cumsum([len(b) for b in inputs], dtype=np.uint64)
    joined = b"".join(inputs)
    data = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, dtype=np.uint8)
    return data, off


def _seed_of(opts):
    s = opts.get("seed")
    if s is None:  # gen_urandom_seed/0 erlamsa_rnd.erl:50-62: three 16-bit values
        raw = os.urandom(6)
        s = tuple(int.from_bytes(raw[2 * i:2 * i + 2], "big") for i in range(3))
    return tuple(int(x) for x in s)


def _configure(eng, opts):
    eng.configure(mutations=actions_to_string(opts.get("mutations")), patterns=actions_to_string(opts.get("patterns")),
                  generators=actions_to_string(opts.get("generators")), blockscale=float(opts.get("blockscale", 1.0)),
                  ssrf_host=opts.get("ssrf_host"), ssrf_port=int(opts.get("ssrf_port", 0)),
                  max_case_bytes=int(opts.get("max_case_bytes", 0)), out_capacity=int(opts.get("out_capacity", 0)),
                  max_slots=int(opts.get("max_slots", 0)), max_case_work=int(opts.get("max_case_work", 0)))


class Unsupported(ValueError):
    """The option map carries keys only erlamsa_main:fuzzer/1 on BEAM can honour: `.keys` (the shim's {error, {unsupported, Keys}})."""

    def __init__(self, keys):
        super().__init__("not served by the GPU path, route this run to erlamsa_main:fuzzer/1: %s" % ", ".join(keys))
        self.keys = keys


def host_only(opts):
    """keys of the Dict that are set and need the BEAM (erlamsa_hip:host_only/1)"""
    keys = [k for k in ("external_mutations", "external_post") if opts.get(k) is not None]
    if opts.get("sequence_muta"):
        keys.append("sequence_muta")
    return keys


def fuzz_batch(inputs, opts=None, return_status=False, device=0):
    """Case I (1-based) of ONE fuzzer/1 run mutates inputs[I-1].  -> list[bytes] (and statuses)."""
    opts = dict(opts or {})
    if host_only(opts):
        raise Unsupported(host_only(opts))
    eng = _engine(device)
    _configure(eng, opts)
    data, off = pack_corpus(list(inputs))
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=_seed_of(opts), first_case=int(opts.get("first_case", 1)), n=len(inputs))
    outs, status = eng.download()
    return (outs, status) if return_status else outs


class EngineLimit(RuntimeError):
    """Cases of a fuzzer() call stopped at an engine limit the reference does not have (statuses 2 overflow,
    3 unsupported, 4 arena full, 5 work budget): `.cases` = [(index, status)], `.results` = what the other cases gave."""

    def __init__(self, cases, results):
        super().__init__("%d case(s) stopped at an engine limit: %s" % (len(cases), cases[:8]))
        self.cases, self.results = cases, results


def fuzzer(opts):
    """erlamsa_main:fuzzer/1 for paths=[direct], output=return: the same input N times.
    Like record_result/2 (erlamsa_main.erl:120-122) empty results are dropped (a crashed worker, status 1, gives <<>>).
    A case that stopped at an engine-only limit is never dropped silently: EngineLimit is raised unless
    opts["on_engine_limit"] == "skip" (then the caller reads the statuses through fuzz_batch(return_status=True))."""
    opts = dict(opts)
    if opts.get("paths", ["direct"]) != ["direct"] or opts.get("output", "return") != "return":
        raise ValueError("only paths=[direct], output=return is served by the GPU path")
    if host_only(opts):
        raise Unsupported(host_only(opts))
    n = int(opts.get("n", 1))
    skip, first = int(opts.get("skip", 0)), int(opts.get("first_case", 1))
    outs, status = fuzz_batch([bytes(opts.get("input", b""))] * n, opts, return_status=True, device=int(opts.get("device", 0)))
    res, limited = [], []
    for i, (o, s) in enumerate(zip(outs, status)):
        if first + i <= skip:                                   # `I =< Skip` (erlamsa_main.erl:191): written to the skip port, which keeps nothing
            continue
        if s == CASE_OK and len(o) > 0:
            res.append(o)
        elif s >= 2:
            limited.append((i, int(s)))
    if limited and opts.get("on_engine_limit", "raise") != "skip":
        raise EngineLimit(limited, res)
    return res


def fuzz(data, opts=None):
    """erlamsa_app:fuzz/1,2: one case; returns the mutated binary ([] when the result is empty,
    like extract_function/1 on an empty list, erlamsa_utils.erl:96-99)."""
    r = fuzzer(dict(opts or {}, input=bytes(data), n=1))
    return r[0] if r else []
