#!/usr/bin/env python3
"""usage: tools/kstats_summary.py <p_kernel_stats.csv> <out.json>
Per-step GEMM time from a `rocprofv3 --kernel-trace --stats` run of bench.py (tools/kstats_bench.sh): the kernels' own
begin..end durations, to set beside bench.py's hipEvent intervals (`roofline.avg_launch_us`).  Steps in the run = calls of
the criterion kernel (one per step); GEMM launches = every m2f_gemm* kernel.  bench.py reads the result from
profiles/kstats_<workload>_<dtype>.json into `roofline.rocprof_avg_launch_us`."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = sum(int(r["Calls"]) for r in rows if "m2f_ce_kernel" in r["Name"])
gemm = [r for r in rows if "m2f_gemm" in r["Name"] or "m2f_mega" in r["Name"]]
calls = sum(int(r["Calls"]) for r in gemm)
total_ns = sum(float(r["TotalDurationNs"]) for r in gemm)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (source_hash: bench.py only quotes this digest while the kernel sources are the ones measured)
out = {"source_hash": bench.source_hash(), "steps_in_run": steps, "gemm_launches_per_step": calls / steps, "gemm_ms_per_step": total_ns / steps / 1e6,
       "avg_launch_us": total_ns / calls / 1e3,
       "kernels": {r["Name"][:100]:
                   {"calls_per_step": int(r["Calls"]) / steps, "avg_us": float(r["AverageNs"]) / 1e3} for r in gemm},
       "source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline (tools/kstats_bench.sh)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
