"""ctypes binding of the CPU oracle (oracle/liberlamsa_oracle.so).

ORACLE = TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package
(erlamsa_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liberlamsa_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle.cpp", "oracle.h", "otp_compat.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _Cfg(C.Structure):
    _fields_ = [("mutations", C.c_char_p), ("patterns", C.c_char_p), ("generators", C.c_char_p),
                ("blockscale", C.c_double), ("ssrf_host", C.c_char_p), ("ssrf_port", C.c_int32),
                ("mode", C.c_int32), ("seed", C.c_int64 * 3), ("first_case", C.c_uint64),
                ("seeds", C.POINTER(C.c_int64)), ("max_case_bytes", C.c_uint64), ("max_case_work", C.c_uint64), ("max_case_seconds", C.c_double),
                ("paths_data", C.c_void_p), ("paths_off", C.c_void_p), ("paths_n", C.c_uint64)]


class _Res(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("off", C.POINTER(C.c_uint64)), ("status", C.POINTER(C.c_int32)),
                ("draws", C.POINTER(C.c_uint64)), ("trace", C.c_char_p), ("trace_len", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.eo_fuzz_batch.restype = C.c_int
        _lib.eo_fuzz_batch.argtypes = [C.POINTER(_Cfg), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(_Res)]
        _lib.eo_last_error.restype = C.c_char_p
        _lib.eo_rand_uniforms.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p]
        _lib.eo_run_mutator.restype = C.c_int32
        _lib.eo_run_mutator.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        _lib.eo_lex_roundtrip.restype = C.c_int32
        _lib.eo_parse_fold.restype = C.c_int32
        _lib.eo_parse_fold.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
        _lib.eo_lex_roundtrip.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.eo_sort_by_priority.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.eo_free.argtypes = [C.c_void_p]
        _lib.eo_free_result.argtypes = [C.POINTER(_Res)]
    return _lib


def uniforms(seed, n):
    out = np.zeros(n, dtype=np.float64)
    lib().eo_rand_uniforms(seed[0], seed[1], seed[2], n, out.ctypes.data)
    return out


def pack(inputs):
    """list[bytes] -> (uint8 data, uint64 off[n+1])"""
    off = np.zeros(len(inputs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(b) for b in inputs], dtype=np.uint64)
    data = np.frombuffer(b"".join(inputs), dtype=np.uint8).copy() if int(off[-1]) else np.zeros(1, dtype=np.uint8)
    return data, off


def fuzz_batch(data, off, seed=(1, 2, 3), mutations=None, patterns=None, generators=None, blockscale=1.0,
               first_case=1, seeds=None, max_case_bytes=0, ssrf_host=None, ssrf_port=0, trace=False, max_case_work=0, max_case_seconds=0.0, paths=None):
    """Returns (list[bytes] outputs, status int32[n], draws uint64[n], trace str|None).
    paths = (data, off) of the corpus the `file` / `jump` generators draw from (default: the batch itself)."""
    n = len(off) - 1
    cfg = _Cfg()
    cfg.mutations = mutations.encode() if mutations is not None else None
    cfg.patterns = patterns.encode() if patterns is not None else None
    cfg.generators = generators.encode() if generators is not None else None
    cfg.blockscale = blockscale
    cfg.ssrf_host = ssrf_host.encode() if ssrf_host else None
    cfg.ssrf_port = ssrf_port
    cfg.first_case = first_case
    cfg.max_case_bytes = max_case_bytes
    cfg.max_case_work = max_case_work
    cfg.max_case_seconds = max_case_seconds
    keep = None
    pkeep = None
    if paths is not None:
        pkeep = (np.ascontiguousarray(paths[0], dtype=np.uint8), np.ascontiguousarray(paths[1], dtype=np.uint64))
        cfg.paths_data, cfg.paths_off, cfg.paths_n = pkeep[0].ctypes.data, pkeep[1].ctypes.data, len(pkeep[1]) - 1
    if seeds is not None:
        keep = np.ascontiguousarray(seeds, dtype=np.int64).reshape(-1)
        assert keep.size == 3 * n
        cfg.mode = 1
        cfg.seeds = keep.ctypes.data_as(C.POINTER(C.c_int64))
    else:
        cfg.mode = 0
        cfg.seed[0], cfg.seed[1], cfg.seed[2] = seed
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    res = _Res()
    rc = lib().eo_fuzz_batch(C.byref(cfg), data.ctypes.data, off.ctypes.data, n, 2 if trace == "full" else 1 if trace else 0, C.byref(res))
    if rc != 0:
        raise RuntimeError("oracle: " + lib().eo_last_error().decode())
    o = np.ctypeslib.as_array(res.off, shape=(n + 1,)).copy()
    total = int(o[-1])
    buf = bytes(np.ctypeslib.as_array(res.data, shape=(max(total, 1),))[:total])
    outs = [buf[int(o[i]):int(o[i + 1])] for i in range(n)]
    status = np.ctypeslib.as_array(res.status, shape=(max(n, 1),))[:n].copy()
    draws = np.ctypeslib.as_array(res.draws, shape=(max(n, 1),))[:n].copy()
    tr = res.trace.decode("latin1") if trace and res.trace else None
    lib().eo_free_result(C.byref(res))
    return outs, status, draws, tr


def run_mutator(name, seed, data):
    """Muta([Bin], []) with worker seed; returns (delta|None on crash, bytes, nblocks)."""
    buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    out = C.POINTER(C.c_uint8)()
    olen = C.c_uint64()
    nb = C.c_uint32()
    d = lib().eo_run_mutator(name.encode(), seed[0], seed[1], seed[2], buf.ctypes.data, len(data), C.byref(out), C.byref(olen), C.byref(nb))
    if d <= -2147483646:
        return None, b"", 0
    b = bytes(np.ctypeslib.as_array(out, shape=(max(olen.value, 1),))[:olen.value])
    lib().eo_free(out)
    return d, b, nb.value


def lex_roundtrip(data):
    buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    out = np.zeros(max(len(data), 1), dtype=np.uint8)
    n = lib().eo_lex_roundtrip(buf.ctypes.data, len(data), out.ctypes.data)
    return n, bytes(out[:len(data)])


def parse_fold(kind, data):
    """kind 'sgml' | 'json': (rc, folded bytes, counts) of the restated parser + serializer."""
    buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    out = C.POINTER(C.c_uint8)()
    olen = C.c_uint64()
    cnt = (C.c_int64 * 3)()
    rc = lib().eo_parse_fold(0 if kind == "sgml" else 1, buf.ctypes.data, len(data), C.byref(out), C.byref(olen), cnt)
    if rc != 0:
        return rc, b"", (0, 0, 0)
    b = bytes(np.ctypeslib.as_array(out, shape=(max(olen.value, 1),))[:olen.value])
    lib().eo_free(out)
    return 0, b, tuple(int(x) for x in cnt)


def sort_by_priority(pris):
    p = np.ascontiguousarray(pris, dtype=np.int32)
    perm = np.zeros(len(p), dtype=np.uint32)
    lib().eo_sort_by_priority(p.ctypes.data, len(p), perm.ctypes.data)
    return perm.tolist()
