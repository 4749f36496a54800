#!/bin/bash
# usage: tools/kcold3.sh  -> kernel-trace average durations of chain GEMM shapes (C3) with the weights L2-warm (pool 1), Infinity-Cache
# resident (pool of ~100 MB) and HBM-cold (pool of > 600 MB): what would prefetching the next launch's weights be worth?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for shape in "1024 1024 1024" "1024 2048 1024" "1024 1024 2048" "1024 768 768"; do
  set -- $shape
  mall=$(( 100000000 / ($2 * $3 * 2) + 1 )); cold=$(( 700000000 / ($2 * $3 * 2) + 1 ))
  for pool in 1 $mall $cold; do
    rm -rf /tmp/kc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python3 $R/tools/gemm_cold.py $shape $pool 400 > /dev/null 2>&1
    python3 - "$shape" $pool <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/kc/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m2f_gemm" in r["Name"]: print("shape", sys.argv[1], "pool", sys.argv[2], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), "calls", r["Calls"], r["Name"][28:80])
PY
  done
done
