#!/bin/bash
# usage: tools/kshapes.sh "M N K" ...   -> kernel-trace avg durations (cache-warm) per shape
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for shape in "$@"; do
  rm -rf /tmp/kc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python3 $R/tools/gemm_cold.py $shape 1 300 > /dev/null 2>&1
  python3 - "$shape" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/kc/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m2f_gemm" in r["Name"]: print("shape", sys.argv[1], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), r["Name"][28:75])
PY
done
