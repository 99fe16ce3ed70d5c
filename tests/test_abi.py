"""The C-ABI shared library loads and exports every symbol include/erlamsa_hip.h declares.
No compute calls here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "erlamsa_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(eh_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    import erlamsa_amd.engine as eng
    lib = eng.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(eng.ABI_SYMBOLS) == names, "engine.ABI_SYMBOLS out of date with the header"
    assert lib.eh_abi_version() == 8


def test_tables_mirror_the_reference():
    import erlamsa_amd as ea
    mt = ea.mutator_table()
    # table order and default priorities of erlamsa_mutations:mutations/1 (erlamsa_mutations.erl:1291-1331)
    assert [m[0] for m in mt] == ("sgm js uw ui ab ad tr2 td num ts1 tr ts2 bd bei bed bf bi ber br sp sr sd snand srnd "
                                  "ld lds lr2 lri lr ls lp lis lrs ft fn fo len b64 uri zip nil").split()
    assert dict((m[0], m[1]) for m in mt)["sgm"] == 10 and dict((m[0], m[1]) for m in mt)["b64"] == 7
    assert sum(m[1] for m in mt) == 66
    pt = ea.pattern_table()
    assert [(p[0], p[1]) for p in pt] == [("od", 1), ("nd", 2), ("bu", 1), ("sk", 2), ("sz", 2), ("cs", 1), ("ar", 1), ("cp", 1), ("co", 0), ("nu", 0)]


def test_no_cpu_fallback_without_a_gpu():
    """Creating a context must fail loudly when no HIP device is present."""
    import erlamsa_amd.engine as eng
    lib = eng.load_library()
    h = ctypes.c_void_p()
    rc = lib.eh_create(0, ctypes.byref(h))
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert rc != 0 and not h.value
        with pytest.raises(eng.EngineError):
            eng.Engine(0)
    else:
        assert rc == 0
        lib.eh_destroy(h)


def test_strerror_and_names():
    import erlamsa_amd.engine as eng
    lib = eng.load_library()
    assert lib.eh_strerror(0) == b"ok"
    assert lib.eh_strerror(-2) == b"no usable HIP device"
    assert lib.eh_kernel_name() == b"eh_mutate_kernel"
    assert lib.eh_mutator_name(999) is None


def test_host_helpers():
    import numpy as np
    import erlamsa_amd as ea
    data, off = ea.pack_corpus([b"ab", b"", b"cde"])
    assert off.tolist() == [0, 2, 2, 5] and bytes(data) == b"abcde"
    assert ea.actions_to_string([("bd", 1), ("num", 3)]) == "bd=1,num=3"
    assert ea.actions_to_string("bd,bf") == "bd,bf"
    from erlamsa_amd import shard
    for n, w in [(10, 3), (65536, 8), (7, 8), (0, 2)]:
        got = [shard.case_range(n, r, w) for r in range(w)]
        assert sum(c for _, c in got) == n
        assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(w - 1))
    assert shard.weak_first_case(0, 0, 8, 100) == 1 and shard.weak_first_case(1, 2, 8, 100) == 1001


def test_erlang_nif_shim_compiles_against_the_header():
    """erlang/c_src/erlamsa_hip_nif.c is the reference-side binding INTEGRATION.md describes.  There is no
    OTP in this image, so it is compile-checked (syntax + types against include/erlamsa_hip.h) with a
    stand-in erl_nif.h that declares only the documented NIF API functions the shim uses."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(root, "tests", "stubs"), "-I", os.path.join(root, "include"),
                        os.path.join(root, "erlang", "c_src", "erlamsa_hip_nif.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_option_keys_only_the_beam_can_honour_are_refused_not_ignored():
    """erlamsa_main:fuzzer/1 honours external_mutations (custom mutator funs appended to the table, erlamsa_main.erl:128),
    external_post (:159) and sequence_muta (:223-235); a batch on the GPU cannot, and must say so BEFORE anything runs - the
    shim's {error, {unsupported, Keys}} (erlang/src/erlamsa_hip.erl host_only/1), api.Unsupported here - so that the caller
    routes the run to the BEAM path.  No engine is needed for a refusal."""
    from erlamsa_amd import api
    for opts, keys in (({"external_mutations": "external_muta"}, ["external_mutations"]),
                       ({"external_post": "external_post"}, ["external_post"]),
                       ({"sequence_muta": True}, ["sequence_muta"]),
                       ({"external_mutations": "m", "external_post": "p", "sequence_muta": 1}, ["external_mutations", "external_post", "sequence_muta"])):
        for call in (lambda o: api.fuzzer(dict(o, input=b"abc", n=2)), lambda o: api.fuzz_batch([b"abc"], o), lambda o: api.fuzz(b"abc", o)):
            with pytest.raises(api.Unsupported) as e:
                call(opts)
            assert e.value.keys == keys
    assert api.host_only({"external_mutations": None, "external_post": None, "sequence_muta": False, "skip": 3}) == []
    # the Erlang shim states the same rule (no OTP in this image: the source is checked, not run)
    erl = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "erlang", "src", "erlamsa_hip.erl")).read()
    assert "host_only(Dict) ->" in erl and "[external_mutations, external_post]" in erl and "{error, {unsupported, Keys}}" in erl
    assert "First + I - 1 > Skip" in erl                                      # skip => N (erlamsa_main.erl:161,191-196)
