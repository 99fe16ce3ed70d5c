#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of one profiling call (gpurun_out/prof_r1, pmc_fetch, pmc_write, pmc_sq)
into the tracked summaries under profiles/.  Usage: tools/collect_profiles.py <round-tag, e.g. r01>"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pat):
    f = sorted(glob.glob(os.path.join(G, pat)))
    if not f:
        sys.exit("missing " + pat)
    return f[-1]


shutil.copy(one("prof_r1/*/*kernel_stats.csv"), os.path.join(P, tag + "_kernel_stats.csv"))
rows = []
hdr = None
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    with open(one(d + "/*/*counter_collection.csv")) as fh:
        rd = csv.reader(fh)
        h = next(rd)
        hdr = hdr or h
        rows += [r for r in rd if "eh_mutate_kernel" in r[h.index("Kernel_Name")]]
with open(os.path.join(P, tag + "_pmc_eh_mutate_kernel.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(hdr)
    w.writerows(rows)


def jline(path):
    with open(path) as fh:
        return [ln for ln in fh if ln.startswith("{")][-1]


with open(os.path.join(P, tag + "_bench_under_rocprof.json"), "w") as fh:
    fh.write(jline(os.path.join(G, "prof_r1_bench.log")))
bl = sorted(glob.glob(os.path.join(G, "bench_[a-z].log")))
if bl:
    with open(os.path.join(P, tag + "_bench.json"), "w") as fh:
        fh.write(jline(bl[-1]))
acc = collections.defaultdict(list)
for r in rows:
    acc[r[hdr.index("Counter_Name")]].append(float(r[hdr.index("Counter_Value")]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
k = [r for r in csv.DictReader(open(os.path.join(P, tag + "_kernel_stats.csv"))) if "eh_mutate_kernel" in r["Name"]][0]
bench = json.loads(jline(os.path.join(G, "prof_r1_bench.log")))
out = {
    "round": tag,
    "command": "rocprofv3 --kernel-trace --stats -f csv -- python bench.py --cpu-sample 0   (defaults: 65536 x 4096 B, 5 steps + 1 warmup, 3 passes in flight)",
    "pmc_command": "rocprofv3 --pmc <COUNTERS> -f csv -- python bench.py --cpu-sample 0 --steps 2 --warmup 1 --inflight 1   (one pass per launch; separate runs for FETCH_SIZE, WRITE_SIZE and the SQ set)",
    "kernel": k["Name"], "calls": int(k["Calls"]), "avg_ms_rocprof": float(k["AverageNs"]) / 1e6,
    "min_ms": float(k["MinNs"]) / 1e6, "max_ms": float(k["MaxNs"]) / 1e6,
    "avg_ms_bench_hip_events_same_run": bench["roofline"]["kernel_ms_avg"],
    "share_of_gpu_time_pct": float(k["Percentage"]),
    "per_launch_counters_mean": mean,
    "FETCH_SIZE_unit": "KiB as reported by rocprofv3; on gfx950 wide coalesced reads are tallied at half their bytes "
                       "(MI355X_MICROARCH.md, HBM section): doubled below; WRITE_SIZE is uncalibrated",
    "traffic_bytes_per_launch": {"fetch_raw": mean["FETCH_SIZE"] * 1024, "fetch_x2": 2 * mean["FETCH_SIZE"] * 1024,
                                 "write": mean["WRITE_SIZE"] * 1024,
                                 "total_fetch_x2_plus_write": (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024},
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "sq_breakdown_of_wave_cycles": {"wait_any(s_waitcnt)": mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"],
                                    "active_inst": mean["SQ_ACTIVE_INST_ANY"] / mean["SQ_WAVE_CYCLES"],
                                    "wait_inst(issue stall)": mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"]},
    "workload_key": {"cases": bench["config"]["cases_per_step_per_gpu"], "size": 4096,
                     "max_case_work": bench["config"]["max_case_work"], "max_case_bytes": bench["config"]["max_case_bytes"],
                     "mutators": bench["config"]["workload"].split("mutators ")[1].split(" (")[0], "patterns": "od,nd,bu"},
}
with open(os.path.join(P, tag + "_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k2: out[k2] for k2 in ("avg_ms_rocprof", "avg_ms_bench_hip_events_same_run", "traffic_bytes_per_launch", "sq_breakdown_of_wave_cycles")}, indent=1))
