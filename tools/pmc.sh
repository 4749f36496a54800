#!/bin/bash
# usage: tools/pmc.sh <tag> <gemm_one.py args...>   (three separate --pmc passes, csv summaries under gpurun_out/pmc_<tag>)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VMEM_RD"
P2="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"
P3="TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM"
P4="TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>$R/gpurun_out/pmc_$tag/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for i in range(1,5):
    files = glob.glob("$R/gpurun_out/pmc_$tag/p%d/**/*counter_collection.csv" % i, recursive=True)
    agg = collections.defaultdict(lambda: [0.0,0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if "gemm" not in row.get("Kernel_Name",""): continue
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    for k,(v,n) in sorted(agg.items()):
        print(f"pass{i} {k:40s} per-dispatch {v/max(n,1):16.1f}  (n={n})")
PY
