"""GPU tests added in round 4 (they sort behind the earlier files): bench.py itself on the real device, the file sink, the ABI-6
additions (context-owned streams, a corpus shared by several contexts, pinned host memory), the engine in a process where torch
initialised the HIP runtime first, and the reference's eunit properties run against the ENGINE."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine_env():
    env = dict(os.environ)
    for k in ("EH_BENCH_CHILD", "EH_BENCH_SIMULATE", "EH_BENCH_DRY", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def test_bench_script_on_the_device():
    """The driver's command with fewer steps: `python bench.py --gpus 1 --steps 2 --warmup 1` as a subprocess on the real device
    (round 3's driver run of it died before its set-up passes and printed nothing).  The line must be BASELINE configs[2] and
    carry roofline, cpu_baseline and a parity count."""
    if util.priming():
        pytest.skip("no oracle cache involved")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--budget-mib", "0"],
                       env=_engine_env(), capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 1000 and d["unit"] == "MB/s" and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["kernel"] == "eh_mutate_kernel" and rf["kernel_ms_avg"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert d["parity_checked"] > 1900                                   # cases 1..2048 against the oracle, bit for bit
    assert "child ended without a result" not in r.stderr and "child killed" not in r.stderr


def test_file_sink_on_the_device(tmp_path):
    """SURVEY §8(f)-3, erlamsa_out.erl:103-123 (`-o "name-%n.ext"`): eh_result_write_files on the GPU - every "%n" of the template
    is the case number (build_name/3), one file per EH_CASE_OK case holding exactly that case's output, other statuses leave
    no file and are counted, an unopenable path is an error (not a crash)."""
    if util.priming():
        pytest.skip("no oracle involved")
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    mat = synth.mixed(300, 1024, seed=41)
    e = ea.Engine(0)
    e.configure(max_case_bytes=1 << 20, big_case_bytes=1 << 20)       # default tables; a small cap so that some cases end as EH_CASE_OVERFLOW
    e.upload_corpus(*synth.as_arena(mat))
    e.fuzz_batch(seed=(9, 8, 7), first_case=101)
    got, st = e.download()
    nf, nb, ns = e.write_files(str(tmp_path / "case-%n-of-%n.bin"), first_number=101, threads=4)
    ok = [i for i in range(len(got)) if st[i] == 0]
    assert len(ok) >= 250 and nf == len(ok) and ns == len(got) - len(ok) and nb == sum(len(got[i]) for i in ok)
    for i in ok:
        with open(tmp_path / ("case-%d-of-%d.bin" % (101 + i, 101 + i)), "rb") as fh:
            assert fh.read() == got[i], i
    assert len(os.listdir(tmp_path)) == nf
    with pytest.raises(ea.EngineError) as ei:
        e.write_files(str(tmp_path / "no-such-dir" / "x-%n"))
    assert ei.value.code == -1
    nf2, _, _ = e.write_files(str(tmp_path / "plain.bin"), threads=1)   # no "%n": every case writes the same name (the reference does too)
    assert nf2 == len(ok) and (tmp_path / "plain.bin").exists()
    e.close()


def test_own_streams_shared_corpus_and_pinned_download():
    """ABI 6: three contexts read ONE uploaded arena (eh_corpus_device + eh_corpus_attach), run different case ranges side by side
    on their own streams (eh_stream) and download into eh_host_alloc memory; every result equals the one a lone context gives
    on the null stream."""
    if util.priming():
        pytest.skip("no oracle involved")
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    from erlamsa_amd.engine import HostBuffer
    mat = synth.mixed(1536, 512, seed=5)
    data, off = synth.as_arena(mat)
    ref = ea.Engine(0)
    ref.configure()
    ref.upload_corpus(data, off)
    ref.fuzz_batch(seed=(3, 1, 4), first_case=1)
    want, wst = ref.download()
    es = [ea.Engine(0) for _ in range(3)]
    for e in es:
        e.configure()
    es[0].upload_corpus(data, off)
    for e in es[1:]:
        e.share_corpus(es[0])
    streams = [e.own_stream() for e in es]
    assert all(streams) and len(set(streams)) == 3
    for k, e in enumerate(es):                                         # all three in flight before the first is collected
        e.fuzz_batch(seed=(3, 1, 4), first_case=512 * k + 1, corpus_first=512 * k, n=512, stream=streams[k])
    for k, e in enumerate(es):
        _, total, _ = e.totals()
        hb = HostBuffer(max(total, 1))
        offk, stk = e.download_into(hb.ptr, hb.size)
        blob = hb.array[:total].tobytes()
        for i in range(512):
            assert int(stk[i]) == int(wst[512 * k + i]) and blob[int(offk[i]):int(offk[i + 1])] == want[512 * k + i], (k, i)
        hb.free()
    for e in es[::-1]:
        e.close()
    ref.close()


def test_engine_in_a_process_where_torch_came_first():
    """The multi-GPU bench imports torch (the RCCL binding) before the engine; no driver-run test did.  A child process: torch
    initialises the HIP runtime and launches kernels, then the engine runs the smoke batch and must give the bytes a torch-free
    process gives."""
    if util.priming():
        pytest.skip("no oracle involved")
    code = r'''
import hashlib, sys
sys.path.insert(0, %r)
if sys.argv[1] == "torch":
    import torch
    x = torch.arange(1 << 20, device="cuda") * 3
    torch.cuda.synchronize()
import erlamsa_amd as ea
from erlamsa_amd import synth
ins = [bytes(r) for r in synth.mixed(256, 1024, seed=77)]
outs, st = ea.fuzz_batch(ins, {"seed": (5, 6, 7)}, return_status=True)
if sys.argv[1] == "torch":
    y = (x + 1).sum().item()
    e = ea.Engine(0); e.configure(); e.upload_corpus(*synth.as_arena(synth.mixed(256, 1024, seed=77)))
    s = torch.cuda.Stream()
    e.fuzz_batch(seed=(5, 6, 7), stream=s.cuda_stream)               # on a torch stream, as INTEGRATION.md says a host may
    o2, _ = e.download()
    assert o2 == outs
h = hashlib.sha1()
for o in outs: h.update(len(o).to_bytes(8, "little")); h.update(o)
print("DIGEST", h.hexdigest(), sum(map(int, st)))
''' % ROOT
    digs = []
    for mode in ("plain", "torch"):
        r = subprocess.run([sys.executable, "-c", code, mode], env=_engine_env(), capture_output=True, text=True, timeout=600)
        ln = [x for x in r.stdout.splitlines() if x.startswith("DIGEST")]
        assert r.returncode == 0 and ln, (mode, r.stdout[-800:], r.stderr[-2000:])
        digs.append(ln[0])
    assert digs[0] == digs[1]


def test_sgml_tokenizer_on_documents_of_many_thousand_events():
    """The tokenizer keeps the 4 096 events around its place of work in LDS (csrc/eh_sgml.h).  Documents far larger than that window:
    pumped ones (a run of elements repeated thousands of times: what the heaviest cases of the bench workload are), one whose first
    tag runs through 12 000 events before it fails and is retried from its next '<' - a jump back across the whole window -, long
    quoted values and comments spanning windows.  Bytes, statuses and draw counts against the oracle, run live."""
    if util.priming():
        pytest.skip("live oracle")
    import erlamsa_amd as ea
    import pyoracle as po
    from erlamsa_amd import synth
    rng = np.random.Generator(np.random.PCG64(44))
    docs = []
    small = synth.sgml_docs(24, seed=8)
    for k in range(10):
        unit = b"".join(small[(3 * k + j) % len(small)] for j in range(3))
        docs.append(b"<doc>" + unit * int(rng.integers(40, 400)) + b"</doc>")
    docs.append(b"<a " + b"x<y " * 4000)                                                   # no '>' anywhere
    docs.append(b"<a " + b"x<y " * 3000 + b"> tail <b k='v'>text</b>")
    docs.append(b"<r>" + b"<e a=\"" + b"v " * 6000 + b"\">t</e>" + b"<!-- " + b"- c " * 5000 + b"-->" + b"</r>")
    docs.append(b"<p>" + b"<q w = 'z' />\n \t" * 5000 + b"</p>")
    docs.append(b"text only < and > and = without a tag " * 3000)
    data, off = po.pack(docs)
    kw = dict(seed=(4, 2, 4), mutations="sgm", patterns="od,nd,bu", max_case_bytes=256 << 20)
    ora = util.oracle_live(data, off, chunk=1, max_case_seconds=60.0, **kw)
    eng = ea.Engine(0)
    eng.configure(mutations="sgm", patterns="od,nd,bu", max_case_bytes=8 << 20, big_case_bytes=512 << 20)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(4, 2, 4))
    got, st = eng.download()
    dr, _ = eng.diag()
    eng.close()
    compared = 0
    for i in range(len(docs)):
        if st[i] in (2, 3) or ora.status[i] in (2, 3, 6):
            continue
        compared += 1
        assert int(st[i]) == int(ora.status[i]) and got[i] == ora.outs[i] and (st[i] != 0 or int(dr[i]) == int(ora.draws[i])), \
            "document %d (%d bytes): first difference at %d" % (i, len(docs[i]), util.first_diff(got[i], ora.outs[i]))
    assert compared >= len(docs) - 2


def test_sgml_tokenizer_replay_of_periodic_documents():
    """csrc/eh_sgml.h replays the tokens of one period of a pumped document instead of walking thousands of copies tag by tag (what
    the heaviest cases of the bench workload spend their time in).  Three ways on pumped documents of 20 KB to a megabyte - period
    starting inside a tag, white space eaten by failed tags, an unterminated quote or comment behind the stretch (nothing may be
    replayed then), two stretches, stretches barely long enough - and, for the tag attempts made one per lane (sg_lane_attempt), documents
    without a period: tag soup, runs of failing attempts, tags of thousands of attributes, names over thousands of '<'.  Four ways:
    replay + lane batches (the default), lane batches only, replay only, the wave-wide machine tag by tag
    (EH_FLAG_SGML_NO_REPLAY | EH_FLAG_SGML_NO_LANES) vs the oracle, run live."""
    if util.priming():
        pytest.skip("live oracle")
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import emu_sgml_replay
    total, bad = emu_sgml_replay.run(n=4, seed=5, scale=2, verbose=True)
    assert total == 60 and bad == 0
    total, bad = emu_sgml_replay.run(n=2, seed=9, scale=12, pats="od", verbose=True)
    assert bad == 0


def test_base64_chunks_decoded_by_the_wave():
    """base64_mutator/2 on text full of base64 (csrc/eh_lex.h b64_decode_wave: one group of four per lane, chunks with white space
    inside packed first): every padding, white space inside groups, between and behind the padding characters, blobs of hundreds of
    kilobytes, hundreds of chunks per block, chunks base64:decode/1 refuses - bytes, statuses and draw counts against the oracle, live."""
    if util.priming():
        pytest.skip("live oracle")
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import emu_b64
    total, bad = emu_b64.run(n=40, seed=3, scale=20, verbose=True)
    assert total == 160 and bad == 0


def test_fuse_paths_agree_on_the_device():
    """erlamsa_fuse:fuse/2 has four routes through the engine (csrc/eh_fuse_lds.h, eh_fuse_red.h in front of eh_fuse.h / eh_fuse2.h).
    The differential scripts the routes were developed with, on the GPU: LDS-resident vs node lists vs oracle on the corner corpora,
    shortened lists vs the lists as they are vs oracle on pumped blocks (and, larger, against each other)."""
    if util.priming():
        pytest.skip("live oracle")
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import emu_fuse_lds
    import emu_fuse_red
    total, bad = emu_fuse_lds.run(n=8, seed=23, verbose=True)
    assert total == 64 and bad == 0
    total, bad = emu_fuse_red.run(n=3, seed=29, verbose=True)
    assert total == 24 and bad == 0
    total, bad = emu_fuse_red.run(n=2, seed=31, big=8, with_oracle=False, verbose=True)
    assert bad == 0
