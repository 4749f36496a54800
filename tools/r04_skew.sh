#!/bin/bash
# table-launch time and fabric reads by start-skew variant (M2F_P8_SKEW): 0 = none, 3000 = within an XCD (default), 536873912 = 2^29 + 3000 = whole XCDs
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_skew; mkdir -p $O
cd /tmp
for SK in 0 3000 536873912 0 3000 536873912; do
  M2F_P8_SKEW=$SK rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$SK -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-leg --repeats 1 --secondary none > $O/b_$SK.json 2> $O/e_$SK.txt
  python3 - $O/ks_$SK $SK <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "p8" in r["Name"] or "adam" in r["Name"]: print("skew", sys.argv[2], r["Name"][:60], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
done
for SK in 0 3000 536873912; do
  M2F_P8_SKEW=$SK rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$SK -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-parity-leg --repeats 1 --secondary none > /dev/null 2> $O/ef_$SK.txt
  python3 - $O/f_$SK $SK <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and "p8" in row["Kernel_Name"]: tot += float(row["Counter_Value"]); n += 1
print("skew", sys.argv[2], "FETCH_SIZE per table launch", round(tot / max(n, 1) / 1e3, 1), "k counts (x 2047 B)", n)
PY
done
