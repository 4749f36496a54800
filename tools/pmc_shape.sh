#!/bin/bash
# usage: TILE=64|128|256 LAYOUT=NT tools/pmc_shape.sh "M N K pool iters"  -> duration + L1/L2 counters of one bf16 GEMM shape
# (separate --pmc passes with --kernel-trace only; gemm_cold.py: pool = 1 cache-warm weights, large pool = cold weights)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
shape="$1"
rm -rf /tmp/kc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python3 $R/tools/gemm_cold.py $shape > /dev/null 2>&1
python3 - "$shape" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/kc/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m2f_gemm" in r["Name"]: print("shape", sys.argv[1], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), r["Name"][28:90])
PY
P1="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES"
P2="TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ"
P3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pm$i -o p -- python3 $R/tools/gemm_cold.py $shape > /dev/null 2>/tmp/pm$i.err || { echo "pass $i failed"; tail -3 /tmp/pm$i.err; }
done
python3 - <<'PY'
import csv, glob, collections
for i in range(1, 4):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/pm%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if "m2f_gemm" not in row.get("Kernel_Name", ""): continue
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(f"  pass{i} {k:32s} per-dispatch {v / max(n, 1):16.1f}  (n={n})")
PY
