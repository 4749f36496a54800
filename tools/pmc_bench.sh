#!/bin/bash
# usage: tools/pmc_bench.sh <kernel-name-substring> [bench args]  -> SQ / TCP / TCC counters per dispatch of that kernel
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
pat="$1"; shift
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES"
P3="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"
P4="TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM"
P5="GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1)); rm -rf /tmp/pb$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pb$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-parity-leg --no-fp32-grad-leg --repeats 1 "$@" > /dev/null 2>/tmp/pb$i.err || { echo "pass $i failed"; tail -3 /tmp/pb$i.err; }
done
python3 - "$pat" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
for i in range(1, 6):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/pb%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""): continue
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(f"pass{i} {k:36s} per-dispatch {v / max(n, 1):16.1f}  (n={n})")
PY
