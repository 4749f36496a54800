#!/bin/bash
# usage: tools/kcold.sh  -> kernel-trace average durations, warm vs HBM-cold weights, for the chain's GEMM shapes
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for shape in "512 768 768" "512 2304 768" "512 2048 768" "512 768 2048" "1024 768 768"; do
  set -- $shape
  cold=$(( 600000000 / ($2 * $3 * 2) + 1 ))
  for pool in 1 $cold; do
    rm -rf /tmp/kc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python3 $R/tools/gemm_cold.py $shape $pool 300 > /dev/null 2>&1
    python3 - "$shape" $pool <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/kc/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m2f_gemm" in r["Name"]: print("shape", sys.argv[1], "pool", sys.argv[2], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), "calls", r["Calls"], r["Name"][28:75])
PY
  done
done
