#!/bin/bash
# usage: tools/kstats_bench.sh <tag> [bench args]  -> rocprofv3 kernel-trace stats of a bench run (in-graph durations)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift
mkdir -p $R/gpurun_out/ks_$tag
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o p -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity-leg --no-fp32-grad-leg --repeats 1 "$@" > $R/gpurun_out/ks_$tag/bench.json 2> $R/gpurun_out/ks_$tag/err.txt
python3 - $R/gpurun_out/ks_$tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:86]:86s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:7.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
