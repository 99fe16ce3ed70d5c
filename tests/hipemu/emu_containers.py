#!/usr/bin/env python3
"""The container paths against the oracle (which calls libz itself): pattern cp on real gzip / zlib inputs - complete, with header
fields, truncated, corrupted, nested, behind a length field - decoded on the device, mutated through the rest of the pattern chain
and compressed again byte for byte (erlamsa_patterns.erl:216-260); pattern ar and mutator zip on real zip archives
(erlamsa_patterns.erl:165-214, erlamsa_mutations.erl:1149-1163).

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_containers.py [cases per configuration]
"""
import io
import os
import sys
import zlib
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea


def _z(data, wbits, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return c.compress(data) + c.flush()


def compressed_corpus(n, seed):
    """gzip / zlib streams of mixed payloads (any encoder settings), and the ways they go wrong"""
    rng = np.random.Generator(np.random.PCG64(seed))
    plain = util.corpus_mixed(n, 3000, seed=seed)
    out = []
    for i, p in enumerate(plain):
        p = p[:int(rng.integers(0, len(p) + 1))]
        kind = i % 15
        lvl = int(rng.choice([0, 1, 6, 9]))
        if kind in (0, 1, 2):
            b = _z(p, 31, lvl)
        elif kind in (3, 4):
            b = _z(p, 15, lvl)
        elif kind == 12:                                                 # OTP >= 20.1: two or three concatenated members are ALL decoded
            q = plain[(i + 1) % len(plain)][:700]
            b = _z(p, 31, lvl) + _z(q, 31, 6) + (_z(p[:50], 31, 1) if rng.random() < 0.5 else b"")
        elif kind == 13:                                                 # ... a member and an unfinished one / stray bytes: data_error, then {compressed, failed}
            g2 = _z(p[::-1], 31, lvl)
            b = _z(p, 31, lvl) + [g2[:len(g2) // 2], b"\x1f", b"\x00\x00", g2[:-3]][int(rng.integers(0, 4))]
        elif kind == 14:                                                 # ... members whose second one has a wrong CRC-32 / ISIZE
            g2 = bytearray(_z(p[:900], 31, lvl)); g2[-1 - int(rng.integers(0, 8))] ^= 0x10
            b = _z(p, 31, lvl) + bytes(g2)
        elif kind == 5:                                                  # header with a name and a comment, trailing bytes behind the member (OTP >= 20.1: data_error)
            raw = _z(p, -15, lvl)
            b = b"\x1f\x8b\x08\x18" + bytes(6) + b"file.bin\x00a comment\x00" + raw + zlib.crc32(p).to_bytes(4, "little") + (len(p) & 0xffffffff).to_bytes(4, "little") + b"TAIL"
        elif kind == 6:                                                  # a zlib stream that just stops: inflate/2 returns what it decoded
            full = _z(p, 15, lvl); b = full[:int(rng.integers(0, len(full) + 1))]
        elif kind == 7:                                                  # a gzip member that just stops: gunzip raises, the zlib attempt too
            full = _z(p, 31, lvl); b = full[:int(rng.integers(0, len(full)))]
        elif kind == 8:                                                  # one flipped bit
            full = bytearray(_z(p, 31 if rng.random() < 0.5 else 15, lvl))
            full[int(rng.integers(0, len(full)))] ^= 1 << int(rng.integers(0, 8)); b = bytes(full)
        elif kind == 9:                                                  # gzip inside gzip
            b = _z(_z(p, 31, lvl), 31, 6)
        elif kind == 10:                                                 # a 16-bit big-endian length field in front of a gzip member (sz then cp)
            g = _z(p[:1500], 31, lvl); b = b"HD" + len(g).to_bytes(2, "big") + g + b"trailer"
        else:
            b = p                                                        # not compressed at all
        out.append(b)
    return out


CP_CONFIGS = [
    # mutators, patterns
    ("bd,bf,bi,sr,sd,num,ld,lr,ab,uw", "cp"),
    ("bd,bf,bi,sr,lr,num,nil=3", "cp,sz,cs,od,nd"),
    (None, "cp=3,sz,sk,od,nd,bu,cs,co,nu"),
]


def run_cp(n=24, big=1 << 25):
    total = skipped = 0
    for ci, (muts, pats) in enumerate(CP_CONFIGS):
        inputs = compressed_corpus(n, seed=300 + ci)
        data, off = po.pack(inputs)
        seed = (21 + ci, 4, 9)
        want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=big, trace="full")
        eng = ea.Engine(0)
        eng.configure(mutations=muts, patterns=pats, max_case_bytes=1 << 20, big_case_bytes=big, flags=ea.engine.EH_FLAG_META_TRACE)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=seed)
        got, gst = eng.download()
        gdr, _ = eng.diag()
        lines = tr.split("\x1e\n")
        bad = []
        for i in range(len(inputs)):
            if gst[i] in (2, 3) or wst[i] in (2, 3):
                skipped += 1
                continue
            ok = got[i] == want[i] and gst[i] == wst[i] and (gst[i] != 0 or gdr[i] == wdr[i])
            if ok and gst[i] == 0:
                ok = util.meta_matches(eng, i, lines[i])
            if not ok:
                bad.append(i)
        eng.close()
        total += len(inputs)
        print("cp config %d: cases %d bad %d, statuses %s" % (ci, len(inputs), len(bad), np.bincount(gst, minlength=4).tolist()), flush=True)
        for i in bad[:3]:
            print("  case %d (input kind %d, %d bytes): first diff %d, len %d vs %d, status %d vs %d, draws %d vs %d, %s" % (
                i, i % 15, len(inputs[i]), util.first_diff(got[i], want[i]), len(got[i]), len(want[i]), gst[i], wst[i], gdr[i], wdr[i], lines[i][:160]))
        assert not bad, "cp config %d: %d cases differ" % (ci, len(bad))
    assert skipped <= total // 10, "%d of %d cases ended at an engine limit" % (skipped, total)
    return total


def _mkzip(files, comment=b"", method=zipfile.ZIP_DEFLATED):
    b = io.BytesIO()
    with zipfile.ZipFile(b, "w", method) as z:
        for name, data in files:
            zi = zipfile.ZipInfo(name, date_time=(1980 + len(name) % 40, 1 + len(data) % 12, 1 + len(name) % 28, len(data) % 24, len(name) % 60, (2 * len(data)) % 60))
            zi.compress_type = method
            z.writestr(zi, data)
        z.comment = comment
    return b.getvalue()


def zip_corpus(n, seed):
    """zip archives as ordinary tools write them (several files, stored and deflated, small and empty files, extensions zip:create
    stores, an archive comment) and the ways they go wrong"""
    rng = np.random.Generator(np.random.PCG64(seed))
    plain = util.corpus_mixed(4 * n, 2500, seed=seed)
    names = ["a.txt", "dir/b.bin", "x/y/z.dat", "notes", "inner.zip", "old.arj", "pack.Z", "d.tar.gz", ".hidden", "UPPER.TXT", "q.zoo", "w.lzh", "e.arc"]
    out = []
    for i in range(n):
        k = int(rng.integers(1, 6))
        files = []
        for j in range(k):
            p = plain[4 * i + j % 4]
            cut = int(rng.choice([0, 3, 9, 10, 11, 200, len(p)]))
            files.append((names[int(rng.integers(0, len(names)))] if j else names[i % len(names)], p[:cut]))
        kind = i % 10
        if kind in (0, 1, 2, 3):
            b = _mkzip(files)
        elif kind == 4:
            b = _mkzip(files, method=zipfile.ZIP_STORED)
        elif kind == 5:
            b = _mkzip(files, comment=b"an archive comment of some length " * int(rng.integers(1, 4)))
        elif kind == 6:                                                  # cut somewhere: no end record, or a central directory that points outside
            full = _mkzip(files); b = full[:int(rng.integers(0, len(full)))]
        elif kind == 7:                                                  # a flipped bit anywhere (header fields, names), or inside the first file's compressed data
            full = bytearray(_mkzip(files))
            lo, hi = 0, len(full)
            u = rng.random()
            if u < 0.4:
                lo = 30 + len(files[0][0]); hi = max(lo + 1, min(len(full), lo + int.from_bytes(full[18:22], "little")))
            elif u < 0.7:                                                # the LAST central-directory entry's signature: the fold fails after the
                lo = bytes(full).rfind(b"PK\x01\x02"); hi = lo + 4       # fun has run (and drawn) for the entries before it
            full[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8)); b = bytes(full)
        elif kind == 8:
            b = _mkzip([])                                               # no entries: the end record alone
        else:
            b = plain[4 * i]                                             # not an archive
        out.append(b)
    return out


ZIP_CONFIGS = [
    ("bd,bf,bi,sr,sd,num,ld,lr,ab,uw", "ar"),
    ("zip", "od,nd"),
    ("zip=5,bd,bf,sr,lr,num", "ar=3,cp,sz,od,nd,bu"),
    (None, "ar=4,cp,sz,sk,od,nd,bu,cs,co,nu"),
]


def run_zip(n=20, big=1 << 25):
    total = skipped = 0
    for ci, (muts, pats) in enumerate(ZIP_CONFIGS):
        inputs = zip_corpus(n, seed=500 + ci)
        data, off = po.pack(inputs)
        seed = (31 + ci, 5, 2)
        want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=big, trace="full")
        eng = ea.Engine(0)
        eng.configure(mutations=muts, patterns=pats, max_case_bytes=1 << 20, big_case_bytes=big, flags=ea.engine.EH_FLAG_META_TRACE)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=seed)
        got, gst = eng.download()
        gdr, _ = eng.diag()
        lines = tr.split("\x1e\n")
        bad = []
        for i in range(len(inputs)):
            if gst[i] == 2 or wst[i] == 2:
                skipped += 1
                continue
            ok = got[i] == want[i] and gst[i] == wst[i] and (gst[i] != 0 or gdr[i] == wdr[i])
            if ok and gst[i] == 0:
                ok = util.meta_matches(eng, i, lines[i])
            if not ok:
                bad.append(i)
        eng.close()
        total += len(inputs)
        print("zip config %d: cases %d bad %d, statuses engine %s oracle %s" % (ci, len(inputs), len(bad), np.bincount(gst, minlength=4).tolist(), np.bincount(wst, minlength=4).tolist()), flush=True)
        for i in bad[:3]:
            print("  case %d (input kind %d, %d bytes): first diff %d, len %d vs %d, status %d vs %d, draws %d vs %d, %s" % (
                i, i % 10, len(inputs[i]), util.first_diff(got[i], want[i]), len(got[i]), len(want[i]), gst[i], wst[i], gdr[i], wdr[i], lines[i][:160]))
        assert not bad, "zip config %d: %d cases differ" % (ci, len(bad))
    assert skipped <= total // 10, "%d of %d cases ended at an engine limit" % (skipped, total)
    return total


def run_budget(n=20, big=1 << 25):
    """The optional work budget (eh_options.max_case_work) counts the codecs' bytes: the engine and the oracle's EngineGuard must stop the
    same cases (status 3) - every other case is byte-identical as before."""
    total = stopped = 0
    for ci, (muts, pats, mk, work) in enumerate([("bd,bf,bi,sr,lr,num,nil=3", "cp,sz,cs,od,nd", compressed_corpus, 300000),
                                                 ("zip=5,bd,bf,sr,lr,num", "ar=3,cp,sz,od,nd,bu", zip_corpus, 700000),
                                                 ("zip", "od,nd", zip_corpus, 120000)]):
        inputs = mk(n, seed=700 + ci)
        data, off = po.pack(inputs)
        seed = (41 + ci, 6, 3)
        want, wst, wdr = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=big, max_case_work=work)[:3]
        eng = ea.Engine(0)
        eng.configure(mutations=muts, patterns=pats, max_case_bytes=1 << 20, big_case_bytes=big, max_case_work=work)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=seed)
        got, gst = eng.download()
        gdr, _ = eng.diag()
        eng.close()
        bad = [i for i in range(len(inputs)) if gst[i] != 2 and wst[i] != 2 and not (got[i] == want[i] and gst[i] == wst[i] and (gst[i] != 0 or gdr[i] == wdr[i]))]
        total += len(inputs); stopped += int((gst == 3).sum())
        print("budget config %d (max_case_work %d): cases %d bad %d, statuses engine %s oracle %s" % (ci, work, len(inputs), len(bad), np.bincount(gst, minlength=4).tolist(), np.bincount(wst, minlength=4).tolist()), flush=True)
        for i in bad[:3]:
            print("  case %d (%d bytes): len %d vs %d, status %d vs %d, draws %d vs %d" % (i, len(inputs[i]), len(got[i]), len(want[i]), gst[i], wst[i], gdr[i], wdr[i]))
        assert not bad, "budget config %d: %d cases differ" % (ci, len(bad))
    assert 0 < stopped < total, "the budgets of this test are meant to stop some cases and not all (%d of %d)" % (stopped, total)
    return total


if __name__ == "__main__":
    assert "emu" in os.environ.get("ERLAMSA_HIP_LIB", ""), "point ERLAMSA_HIP_LIB at the emulator build"
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    print("containers ok: cp %d cases, zip %d cases, with a work budget %d cases" % (run_cp(n), run_zip(n), run_budget(min(n, 20))))
