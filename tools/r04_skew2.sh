#!/bin/bash
# table-launch time by start-skew variant (M2F_P8_SKEW) on the final round-4 epilogue: 0 = none; 2^29 + c = whole XCDs against each other, c cycles per k-tile assumed;
# + 2^30 = workgroups without slack too.  Two passes over the list (box drift).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_skew2; mkdir -p $O
cd /tmp
for pass in 1 2; do
for SK in 0 536872912 536873912 536875412 536876912 3000 1610615736; do
  M2F_P8_SKEW=$SK rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${SK}_$pass -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-leg --no-fp32-grad-leg --repeats 1 --secondary none > $O/b_$SK.json 2> $O/e_$SK.txt
  python3 - $O/ks_${SK}_$pass $SK <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "p8" in r["Name"]: print("skew", sys.argv[2], "table launch avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), flush=True)
PY
done
done
