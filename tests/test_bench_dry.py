"""bench.py itself, end to end, on the CPU wavefront emulator (ERLAMSA_HIP_LIB = the emulator build; a single-GPU run uses no torch:
corpus through eh_corpus_upload, the contexts' own streams - the null stream on the emulator): the step loop, the roofline arithmetic, the parity leg against the oracle (the bench fails unless its sample agrees), the
PCIe and work-budget legs, the JSON line - for the driver's configuration shape and for `--config 5` (counter-hash corpus, generator jump over the whole arena, strong scaling).  Sizes are tiny; the numbers mean nothing, the code paths do."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
SMALL = ["--steps", "2", "--warmup", "1", "--inflight", "2", "--max-slots", "4", "--out-gib", "1", "--pool-gib", "1", "--case-mib", "1", "--big-mib", "32",
         "--cpu-sample", "8", "--cpu-threads", "2", "--setup-seconds", "0"]


def _run(extra):
    import build_emu
    env = dict(os.environ, EH_BENCH_DRY="1", ERLAMSA_HIP_LIB=build_emu.build())
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads(lines[0])


def test_bench_script_runs_end_to_end_on_the_emulator():
    d = _run(["--cases", "16", "--size", "256", "--mutations", "bd,bf,bi,sr,num,lr,ab", "--budget-mib", "1", "--pcie", "1"])
    assert d["metric"] == "mutated_MB_per_s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["cases_per_step_per_gpu"] == 16
    # --cpu-sample 8 = rows 0..3 of THREE passes of the run (the first, one in the middle, the last: bench.parity_windows)
    assert d["case_status"]["ok"] == 32 and d["parity_checked"] + d["parity"]["not_compared"] == 12 and d["parity_checked"] >= 11
    assert [c[1] - c[0] + 1 for c in d["parity"]["cases"]] == [4, 4, 4] and d["parity"]["cases"][0][0] == 1 and d["parity"]["cases"][2][0] == 2 * 16 + 1
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["algorithmic_bytes_per_launch"] > 16 * 256
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 2 and "pcie" in d and "with_work_budget" in d
    hl, pool = d["host_loop_ms_per_step"], d["config"]["work_area_pool"]
    assert set(hl) == {"collect", "collect_max", "on_result", "launch", "launch_max", "no_context_free"} and all(v >= 0 for v in hl.values())
    assert len(pool["peak_wanted"]) == len(pool["areas"]) and all(a >= 1 for a in pool["areas"])


def test_bench_config_5_shape_runs_on_the_emulator():
    d = _run(["--config", "5", "--cases", "12", "--size", "4096", "--budget-mib", "0", "--pcie", "0", "--steps", "1", "--warmup", "0", "--inflight", "1", "--cpu-sample", "6"])
    assert "configs[4]" in d["config"]["workload"] and "generator jump" in d["config"]["workload"] and d["scaling"] == "strong"
    assert d["case_status"]["ok"] == 12 and d["parity_checked"] == 6        # the oracle's Paths are the whole counter-hash arena


@pytest.mark.parametrize("scaling", ["weak", "strong", "weak-without-rccl"])
def test_bench_script_two_ranks_over_gloo_on_the_emulator(scaling):
    """bench.py's OWN N > 1 path (process group, arena generated on rank 0 and broadcast, the checksum agreement of all ranks, one
    context attaching the broadcast tensors and the others sharing it, step loop per rank, barrier, MAX / SUM reduction, one JSON
    line from rank 0) as two CPU ranks over gloo with the emulator build of the engine - everything of the driver's multi-GPU run
    except RCCL itself.  Launched the way the driver launches it."""
    import build_emu
    import test_comm_abi
    test_comm_abi._build()                                                  # build/libfake_rccl.so: the stand-in behind EH_RCCL_LIB
    env = dict(os.environ, EH_BENCH_BACKEND="gloo", ERLAMSA_HIP_LIB=build_emu.build(), EH_RCCL_LIB=test_comm_abi.FAKE)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EH_BENCH_CHILD"):
        env.pop(k, None)
    no_rccl = scaling == "weak-without-rccl"      # the library cannot load RCCL: every rank agrees to load the arena over torch.distributed instead
    if no_rccl:
        env["EH_RCCL_LIB"] = os.path.join(ROOT, "build", "no_such_rccl.so")
        scaling = "weak"
    port = 29700 + (os.getpid() % 200) + (2 if no_rccl else 0 if scaling == "weak" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", scaling, "--cases", "16", "--size", "256", "--mutations", "bd,bf,bi,sr,num,lr,ab"] + SMALL
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["world_size_seen_by_torch_distributed"] == 2 and d["config"]["arena_equal_on_all_ranks"] is True
    if scaling == "weak":
        assert d["strong_scaling_leg"]["cases_per_step_all_ranks"] == 16 and d["strong_scaling_leg"]["cases_per_s"] > 0     # (value is MB/s to one decimal: 16 cases of the emulator can round to 0.0)
    assert ("fall-back" if no_rccl else "RCCL inside the library") in d["config"]["arena_transport"]
    assert d["case_status"]["ok"] == (32 if scaling == "weak" else 16)          # rank 0's share: weak = its own 16 cases per step, strong = half of one run's
