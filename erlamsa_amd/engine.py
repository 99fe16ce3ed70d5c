"""ctypes binding of liberlamsa_hip.so (C ABI declared in include/erlamsa_hip.h).

This is the host-side plumbing the tests and bench use; the Erlang NIF shim in
erlang/ binds the very same symbols.  The library is REQUIRED: there is no CPU
fallback, and loading fails loudly when the HIP extension has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ERLAMSA_HIP_LIB") or os.path.join(_HERE, "liberlamsa_hip.so")

EH_ABI_VERSION = 8
EH_FLAG_ORDERED_OUTPUT = 1
EH_FLAG_META_TRACE = 2
EH_FLAG_FUSE_NO_LDS = 4
EH_FLAG_FUSE_NO_REDUCE = 8
EH_FLAG_SGML_NO_REPLAY = 16
EH_FLAG_SGML_NO_LANES = 32
EH_FLAG_NO_COOP = 64

CASE_OK, CASE_CRASHED, CASE_OVERFLOW, CASE_UNSUPPORTED, CASE_ARENA_FULL, CASE_BUDGET = 0, 1, 2, 3, 4, 5

# every symbol include/erlamsa_hip.h declares
ABI_SYMBOLS = [
    "eh_create", "eh_destroy", "eh_configure", "eh_corpus_upload", "eh_corpus_attach", "eh_fuzz_batch",
    "eh_fuzz_calls", "eh_reserve", "eh_sync", "eh_result_device", "eh_result_download", "eh_result_fetch", "eh_result_totals", "eh_result_summary", "eh_result_occupancy", "eh_result_diag", "eh_result_cycles", "eh_result_peak", "eh_result_meta", "eh_result_write_files", "eh_result_prof", "eh_selftest_movers", "eh_selftest_zlib",
    "eh_last_kernel_ms", "eh_pool_stats", "eh_coop_stats", "eh_kernel_name", "eh_abi_version", "eh_mutator_count", "eh_mutator_name",
    "eh_mutator_default_pri", "eh_mutator_on_gpu", "eh_pattern_count", "eh_pattern_name",
    "eh_pattern_default_pri", "eh_pattern_on_gpu", "eh_strerror", "eh_last_error",
    "eh_coalesce_limits", "eh_submit", "eh_flush", "eh_poll", "eh_cancel",
    "eh_corpus_device", "eh_stream", "eh_host_alloc", "eh_host_free", "eh_selftest_sort_by_priority", "eh_last_error_copy",
    "eh_comm_unique_id", "eh_comm_init", "eh_comm_init_local", "eh_comm_destroy", "eh_corpus_broadcast", "eh_corpus_allgather",
    "eh_corpus_broadcast_local", "eh_device_count", "eh_meta_atom_count", "eh_meta_atom_name", "eh_batch_done",
]


class EhOptions(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("mutations", C.c_char_p), ("patterns", C.c_char_p),
                ("generators", C.c_char_p), ("blockscale", C.c_double), ("ssrf_host", C.c_char_p),
                ("ssrf_port", C.c_int32), ("max_case_bytes", C.c_uint64), ("out_capacity", C.c_uint64),
                ("max_case_work", C.c_uint64), ("max_slots", C.c_uint32), ("flags", C.c_uint32), ("big_case_bytes", C.c_uint64),
                ("pool_bytes", C.c_uint64), ("download_chunk_bytes", C.c_uint64), ("fuse_stream_min", C.c_uint64),
                ("sequence_muta", C.c_uint32), ("reserved0", C.c_uint32)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("erlamsa_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """Loads liberlamsa_hip.so; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime: when both live in one process torch must initialise first (the
    # engine then shares that runtime; the other order leaves torch without devices — INTEGRATION.md §4)
    import sys
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64p, i64p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    lib.eh_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.eh_destroy.argtypes = [vp]
    lib.eh_destroy.restype = None
    lib.eh_configure.argtypes = [vp, C.POINTER(EhOptions)]
    lib.eh_corpus_upload.argtypes = [vp, vp, vp, C.c_uint64]
    lib.eh_corpus_attach.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
    lib.eh_fuzz_batch.argtypes = [vp, i64p, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    lib.eh_fuzz_calls.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp]
    lib.eh_sync.argtypes = [vp]
    lib.eh_batch_done.argtypes = [vp, C.POINTER(C.c_int)]
    lib.eh_reserve.argtypes = [vp, C.c_uint64]
    lib.eh_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p]
    lib.eh_result_download.argtypes = [vp, vp, C.c_uint64, vp, vp]
    lib.eh_result_totals.argtypes = [vp, u64p, u64p, u64p]
    lib.eh_result_fetch.argtypes = [vp, C.c_uint64, vp, C.c_uint64, u64p]
    lib.eh_result_diag.argtypes = [vp, vp, vp]
    lib.eh_result_cycles.argtypes = [vp, vp]
    lib.eh_result_peak.argtypes = [vp, vp]
    lib.eh_result_write_files.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_uint32, u64p, u64p, u64p]
    lib.eh_result_meta.argtypes = [vp, C.c_uint64, vp, C.c_uint64, u64p]
    lib.eh_result_prof.argtypes = [vp, vp]
    lib.eh_selftest_movers.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    lib.eh_selftest_zlib.argtypes = [vp, C.c_int, vp, C.c_uint64, vp, C.c_uint64, vp, vp]
    lib.eh_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.eh_pool_stats.argtypes = [vp, vp]
    lib.eh_result_summary.argtypes = [vp, vp]
    lib.eh_result_occupancy.argtypes = [vp, vp]
    lib.eh_coop_stats.argtypes = [vp, vp]
    lib.eh_coalesce_limits.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.eh_submit.argtypes = [vp, vp, C.c_uint64, i64p, u64p]
    lib.eh_flush.argtypes = [vp]
    lib.eh_meta_atom_name.restype = C.c_char_p
    lib.eh_meta_atom_name.argtypes = [C.c_int]
    lib.eh_comm_unique_id.argtypes = [vp]
    lib.eh_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.eh_comm_init_local.argtypes = [vp, C.c_int]
    lib.eh_comm_destroy.argtypes = [vp]
    lib.eh_corpus_broadcast.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    lib.eh_corpus_allgather.argtypes = [vp, vp, vp, C.c_uint64]
    lib.eh_corpus_broadcast_local.argtypes = [vp, C.c_int, C.c_int]
    lib.eh_cancel.argtypes = [vp, C.c_uint64]
    lib.eh_poll.argtypes = [vp, C.c_uint64, vp, C.c_uint64, u64p, C.POINTER(C.c_int32)]
    lib.eh_corpus_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), u64p, u64p]
    lib.eh_stream.argtypes = [vp, C.POINTER(vp)]
    lib.eh_host_alloc.argtypes = [C.POINTER(vp), C.c_uint64]
    lib.eh_host_free.argtypes = [vp]
    lib.eh_host_free.restype = None
    lib.eh_selftest_sort_by_priority.argtypes = [vp, C.c_uint32, vp]
    lib.eh_kernel_name.restype = C.c_char_p
    lib.eh_abi_version.restype = C.c_uint32
    for f in ("eh_mutator_name", "eh_pattern_name", "eh_strerror"):
        getattr(lib, f).restype = C.c_char_p
        getattr(lib, f).argtypes = [C.c_int]
    lib.eh_last_error.restype = C.c_char_p
    lib.eh_last_error.argtypes = [vp]
    _lib = lib
    return lib


def mutator_table():
    lib = load_library()
    return [(lib.eh_mutator_name(i).decode(), lib.eh_mutator_default_pri(i), bool(lib.eh_mutator_on_gpu(i)))
            for i in range(lib.eh_mutator_count())]


def pattern_table():
    lib = load_library()
    return [(lib.eh_pattern_name(i).decode(), lib.eh_pattern_default_pri(i), bool(lib.eh_pattern_on_gpu(i)))
            for i in range(lib.eh_pattern_count())]


def sort_by_priority(pris):
    """the engine's host-side erlamsa_utils:sort_by_priority/1 (a self-test hook): positions of `pris` in sorted order"""
    p = np.ascontiguousarray(pris, dtype=np.uint32)
    perm = np.zeros(max(len(p), 1), dtype=np.uint32)
    rc = load_library().eh_selftest_sort_by_priority(p.ctypes.data, len(p), perm.ctypes.data)
    if rc != 0:
        raise EngineError(rc, "eh_selftest_sort_by_priority")
    return [int(x) for x in perm[:len(p)]]


def gpu_mutators():
    return [n for n, _, g in mutator_table() if g]


def gpu_patterns():
    return [n for n, _, g in pattern_table() if g]


class Engine:
    """One engine context on one GPU."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.eh_create(device, C.byref(h))
        if rc != 0:
            raise EngineError(rc, self.lib.eh_strerror(rc).decode())
        self.h = h
        self._keep = []
        self.n_corpus = 0

    def close(self):
        if self.h:
            self.lib.eh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, self.lib.eh_last_error(self.h).decode() or self.lib.eh_strerror(rc).decode())

    def configure(self, mutations=None, patterns=None, generators=None, blockscale=1.0, ssrf_host=None, ssrf_port=0,
                  max_case_bytes=0, out_capacity=0, max_slots=0, flags=0, max_case_work=0, big_case_bytes=0,
                  pool_bytes=0, download_chunk_bytes=0, fuse_stream_min=0, sequence_muta=False):
        o = EhOptions()
        o.sequence_muta = 1 if sequence_muta else 0
        o.abi_version = EH_ABI_VERSION
        o.mutations = mutations.encode() if mutations is not None else None
        o.patterns = patterns.encode() if patterns is not None else None
        o.generators = generators.encode() if generators is not None else None
        o.blockscale = blockscale
        o.ssrf_host = ssrf_host.encode() if ssrf_host else None
        o.ssrf_port = ssrf_port
        o.max_case_bytes = max_case_bytes
        o.big_case_bytes = big_case_bytes
        o.pool_bytes, o.download_chunk_bytes, o.fuse_stream_min = pool_bytes, download_chunk_bytes, fuse_stream_min
        o.out_capacity = out_capacity
        o.max_slots = max_slots
        o.max_case_work = max_case_work
        o.flags = flags
        self._chk(self.lib.eh_configure(self.h, C.byref(o)))

    def upload_corpus(self, data, off):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        self._chk(self.lib.eh_corpus_upload(self.h, data.ctypes.data, off.ctypes.data, n))
        self.n_corpus = n

    def attach_corpus(self, d_data_ptr, d_off_ptr, n, nbytes):
        """d_*_ptr: raw device addresses (e.g. torch tensor .data_ptr())."""
        self._chk(self.lib.eh_corpus_attach(self.h, C.c_void_p(d_data_ptr), C.c_void_p(d_off_ptr), n, nbytes))
        self.n_corpus = n

    # ---- multi-GPU: the arena over RCCL, called from inside the library (include/erlamsa_hip.h, csrc/eh_comm.h)
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128 bytes every rank hands to comm_init (eh_comm_unique_id = ncclGetUniqueId)"""
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.eh_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(rc, lib.eh_strerror(rc).decode() + " (RCCL could not be loaded: EH_RCCL_LIB)")
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks):
        """collective over the nranks processes (one per GPU): ncclCommInitRank on this context's device"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(self.lib.eh_comm_init(self.h, buf, rank, nranks))

    def comm_destroy(self):
        self._chk(self.lib.eh_comm_destroy(self.h))

    def corpus_broadcast(self, root, data=None, off=None):
        """root passes the arena, the others nothing; afterwards every rank's context holds it (eh_corpus_broadcast)"""
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            off = np.ascontiguousarray(off, dtype=np.uint64)
            self._chk(self.lib.eh_corpus_broadcast(self.h, root, data.ctypes.data, off.ctypes.data, len(off) - 1))
        else:
            self._chk(self.lib.eh_corpus_broadcast(self.h, root, None, None, 0))
        self.n_corpus = self.corpus_device()[2]

    def corpus_allgather(self, data, off):
        """every rank passes its shard (same entries and bytes on all ranks); afterwards every context holds all shards in rank order"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self._chk(self.lib.eh_corpus_allgather(self.h, data.ctypes.data, off.ctypes.data, len(off) - 1))
        self.n_corpus = self.corpus_device()[2]

    @staticmethod
    def comm_init_local(engines):
        """one process, several devices (one context each): ncclCommInitAll"""
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        rc = engines[0].lib.eh_comm_init_local(arr, len(engines))
        engines[0]._chk(rc)

    @staticmethod
    def corpus_broadcast_local(engines, root=0):
        """the corpus loaded on engines[root] goes to the devices of all the others (eh_corpus_broadcast_local)"""
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        rc = engines[0].lib.eh_corpus_broadcast_local(arr, len(engines), root)
        engines[root]._chk(rc)
        for e in engines:
            e.n_corpus = engines[root].n_corpus

    def corpus_device(self):
        """-> (d_data address, d_off address, n, nbytes) of the loaded corpus (eh_corpus_device)"""
        d, o, n, nb = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.eh_corpus_device(self.h, C.byref(d), C.byref(o), C.byref(n), C.byref(nb)))
        return d.value, o.value, n.value, nb.value

    def share_corpus(self, other):
        """this context reads the corpus `other` holds (same device); `other` must outlive it"""
        d, o, n, nb = other.corpus_device()
        self.attach_corpus(d, o, n, nb)
        self._keep.append(other)

    def own_stream(self):
        """raw handle of the context's own HIP stream (eh_stream); 0 on the CPU emulator (the null stream)"""
        s = C.c_void_p()
        self._chk(self.lib.eh_stream(self.h, C.byref(s)))
        return s.value or 0

    def fuzz_batch(self, seed=(1, 2, 3), first_case=1, corpus_first=0, n=None, stream=0):
        if n is None:
            n = self.n_corpus - corpus_first
        s = (C.c_int64 * 3)(*seed)
        self._chk(self.lib.eh_fuzz_batch(self.h, s, first_case, corpus_first, n, C.c_void_p(stream)))
        self.last_n = n

    def fuzz_calls(self, seeds, corpus_first=0, stream=0):
        seeds = np.ascontiguousarray(seeds, dtype=np.int64).reshape(-1)
        n = seeds.size // 3
        self._chk(self.lib.eh_fuzz_calls(self.h, seeds.ctypes.data, corpus_first, n, C.c_void_p(stream)))
        self.last_n = n

    def reserve(self, max_cases):
        self._chk(self.lib.eh_reserve(self.h, max_cases))

    def sync(self):
        self._chk(self.lib.eh_sync(self.h))

    def done(self):
        """has the last batch finished?  (eh_batch_done: never blocks)"""
        d = C.c_int()
        self._chk(self.lib.eh_batch_done(self.h, C.byref(d)))
        return bool(d.value)

    def totals(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.eh_result_totals(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def summary(self):
        """(input bytes, output bytes, cases, int64[6] cases by status) of the last batch, summed on the device (eh_result_summary)"""
        v = np.zeros(9, dtype=np.uint64)
        self._chk(self.lib.eh_result_summary(self.h, v.ctypes.data_as(C.c_void_p)))
        return int(v[0]), int(v[1]), int(v[2]), v[3:9].astype(np.int64)

    def occupancy(self):
        """(workgroup lifetimes, of that in cases, lingering for posted chunks: 100 MHz ticks summed over the batch's workgroups; workgroups; wave slots of the device) - eh_result_occupancy"""
        v = np.zeros(5, dtype=np.uint64)
        self._chk(self.lib.eh_result_occupancy(self.h, v.ctypes.data_as(C.c_void_p)))
        return tuple(int(x) for x in v)

    def kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.eh_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def pool_stats(self):
        """Work-area pool of the device (eh_pool_stats): per-tier (1..) area bytes, areas, takes, waits."""
        v = np.zeros(64, dtype=np.uint64)
        self._chk(self.lib.eh_pool_stats(self.h, v.ctypes.data_as(C.c_void_p)))
        nt = int(v[40])
        ts = range(1, nt + 1)
        return {"slot_bytes": int(v[51]), "slots": int(v[62]), "area_bytes": [int(v[51 + t]) for t in ts], "areas": [int(v[41 + t]) & 0xFFFFFFFF for t in ts],
                "peak_wanted": [int(v[41 + t]) >> 32 for t in ts],
                "taken": [int(v[2 * t]) for t in ts], "waits": [int(v[30 + t]) for t in ts], "wait_ticks": [int(v[20 + t]) for t in ts],
                "contexts": int(v[61])}

    def coop_stats(self):
        """Cooperative execution of heavy cases on this device (eh_coop_stats)."""
        v = np.zeros(8, dtype=np.uint64)
        self._chk(self.lib.eh_coop_stats(self.h, v.ctypes.data_as(C.c_void_p)))
        return {"loops_posted": int(v[0]), "chunks_by_helpers": int(v[1]), "chunks_by_posters": int(v[2]), "poster_wait_cycles": int(v[3]), "board_full": int(v[4])}

    def download(self):
        """-> (list[bytes] per case, status int32[n])"""
        n = self.last_n
        _, total, _ = self.totals()
        data = np.zeros(max(total, 1), dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.lib.eh_result_download(self.h, data.ctypes.data, data.size, off.ctypes.data, status.ctypes.data))
        buf = data.tobytes()
        outs = [buf[int(off[i]):int(off[i + 1])] for i in range(n)]
        return outs, status[:n]

    def lens(self):
        """output length of every case of the last batch, uint64[n] (no bytes are copied)"""
        n = self.last_n
        off = np.zeros(n + 1, dtype=np.uint64)
        self._chk(self.lib.eh_result_download(self.h, None, 0, off.ctypes.data, None))
        return np.diff(off)

    def fetch(self, i, length=None):
        """output bytes of case i of the last batch (eh_result_fetch)"""
        ln = C.c_uint64()
        if length is None:
            rc = self.lib.eh_result_fetch(self.h, i, None, 0, C.byref(ln))
            if rc == 0:
                return b""
            length = ln.value
        buf = np.zeros(max(int(length), 1), dtype=np.uint8)
        self._chk(self.lib.eh_result_fetch(self.h, i, buf.ctypes.data, int(length), C.byref(ln)))
        return buf[:ln.value].tobytes()

    def download_into(self, host_ptr, cap):
        """Case-ordered outputs of the last batch into caller memory at `host_ptr` (`cap` bytes; pinned or registered
        host memory gets the full PCIe rate).  -> (off uint64[n+1], status int32[n])"""
        n = self.last_n
        off = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.lib.eh_result_download(self.h, C.c_void_p(host_ptr), cap, off.ctypes.data, status.ctypes.data))
        return off, status[:n]

    def status(self):
        n = self.last_n
        st = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.lib.eh_result_download(self.h, None, 0, None, st.ctypes.data))
        return st[:n]

    def diag(self):
        n = self.last_n
        draws = np.zeros(max(n, 1), dtype=np.uint64)
        lastm = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.lib.eh_result_diag(self.h, draws.ctypes.data, lastm.ctypes.data))
        return draws[:n], lastm[:n]

    def cycles(self):
        n = self.last_n
        cyc = np.zeros(max(n, 1), dtype=np.uint64)
        self._chk(self.lib.eh_result_cycles(self.h, cyc.ctypes.data))
        return cyc[:n]

    def write_files(self, template, first_number=1, threads=0):
        """erlamsa_out's file sink: every EH_CASE_OK case of the last batch -> template with "%n" = case number.
        -> (files written, bytes written, cases without a file)"""
        f, b, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.eh_result_write_files(self.h, template.encode(), first_number, threads, C.byref(f), C.byref(b), C.byref(s)))
        return f.value, b.value, s.value

    def meta_raw(self, i):
        """event bytes of case i's meta trace (eh_result_meta; configure with flags=EH_FLAG_META_TRACE)"""
        buf = (C.c_uint8 * 32768)()
        n = C.c_uint64()
        self._chk(self.lib.eh_result_meta(self.h, i, buf, 32768, C.byref(n)))
        return bytes(buf[:min(n.value, 32768)])

    def meta_atoms(self):
        return [self.lib.eh_meta_atom_name(k).decode() for k in range(self.lib.eh_meta_atom_count())]

    def meta_terms(self, i):
        """-> (terms, truncated): the reference's Meta list of case i, element by element, in the order erlamsa_main.erl:58-70 prints it
        (erlamsa_amd/meta.py: render / lines give the ~p text)"""
        from . import meta as _meta
        return _meta.decode(self.meta_raw(i), self.meta_atoms())

    def meta(self, i):
        """The short form: list of (kind, name), kind in 'failed', 'used', 'pattern', 'skipped_big' (mutator / pattern codes), in the
        order the reference makes the entries."""
        from . import meta as _meta
        terms, cut = self.meta_terms(i)
        out = _meta.legacy(terms, [self.lib.eh_mutator_name(k).decode() for k in range(self.lib.eh_mutator_count())])
        return out + ([("truncated", "")] if cut else [])

    def peak(self):
        """Per-case high-water mark of work memory (bytes)."""
        n = self.last_n
        pk = np.zeros(max(n, 1), dtype=np.uint64)
        self._chk(self.lib.eh_result_peak(self.h, pk.ctypes.data))
        return pk[:n]

    def prof(self):
        pr = np.zeros(256, dtype=np.uint64)
        self._chk(self.lib.eh_result_prof(self.h, pr.ctypes.data))
        return pr

    def selftest_zlib(self, op, data, cap=None):
        """device deflate / inflate (csrc/eh_zlib.h): op 0 raw deflate, 1 gzip, 2 zlib, 4 gunzip, 5 zlib inflate -> bytes or None where the reference raises"""
        src = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        if cap is None:
            cap = len(data) + len(data) // 8 + 1024 if op <= 2 else max(64 * len(data), 1 << 20)
        out = np.zeros(cap + 16, dtype=np.uint8)
        n = C.c_uint64(0)
        ok = C.c_int32(0)
        self._chk(self.lib.eh_selftest_zlib(self.h, op, src.ctypes.data, len(data), out.ctypes.data, cap, C.byref(n), C.byref(ok)))
        return bytes(out[:n.value]) if ok.value else None

    def selftest_movers(self, buf, jobs):
        buf = np.ascontiguousarray(buf, dtype=np.uint8).copy()
        jobs = np.ascontiguousarray(jobs, dtype=np.uint32)
        eq = np.zeros(len(jobs), dtype=np.uint32)
        self._chk(self.lib.eh_selftest_movers(self.h, buf.ctypes.data, buf.size, jobs.ctypes.data, len(jobs), eq.ctypes.data))
        return buf, eq

    # ---- request coalescing (eh_submit / eh_flush / eh_poll)
    def coalesce_limits(self, flush_cases, flush_bytes):
        self._chk(self.lib.eh_coalesce_limits(self.h, flush_cases, flush_bytes))

    def submit(self, data, seed):
        """one erlamsa_app:fuzz(Bin, #{seed => Seed}) request -> ticket"""
        b = bytes(data)
        t = C.c_uint64()
        buf = (C.c_char * max(len(b), 1)).from_buffer_copy(b or b"\0")
        self._chk(self.lib.eh_submit(self.h, C.cast(buf, C.c_void_p), len(b), (C.c_int64 * 3)(*seed), C.byref(t)))
        return t.value

    def flush(self):
        self._chk(self.lib.eh_flush(self.h))

    def cancel(self, ticket):
        """Gives a ticket up (its result is dropped / freed)."""
        self._chk(self.lib.eh_cancel(self.h, ticket))

    def poll(self, ticket, cap=1 << 16):
        """-> (status, bytes), or None while the request has not been launched (EH_E_AGAIN)"""
        while True:
            out = (C.c_uint8 * cap)()
            n, st = C.c_uint64(), C.c_int32()
            rc = self.lib.eh_poll(self.h, ticket, out, cap, C.byref(n), C.byref(st))
            if rc == -7:
                return None
            if rc == -1 and n.value > cap:
                cap = int(n.value)
                continue
            self._chk(rc)
            return st.value, bytes(out[:n.value])

    def result_device(self):
        d, o, l, s = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        tot = C.c_uint64()
        self._chk(self.lib.eh_result_device(self.h, C.byref(d), C.byref(o), C.byref(l), C.byref(s), C.byref(tot)))
        return d.value, o.value, l.value, s.value, tot.value


class HostBuffer:
    """page-locked host memory from the library (eh_host_alloc): .ptr, .size, .array (uint8 numpy view)"""

    def __init__(self, size):
        self.lib = load_library()
        p = C.c_void_p()
        rc = self.lib.eh_host_alloc(C.byref(p), size)
        if rc != 0:
            raise EngineError(rc, self.lib.eh_strerror(rc).decode())
        self.ptr, self.size = p.value, size
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(size, 1),))

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.eh_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
