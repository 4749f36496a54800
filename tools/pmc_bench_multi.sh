#!/bin/bash
# usage: tools/pmc_bench_multi.sh <out-dir> <bench args...> -- <name=kernel-substring> ...
# SQ / TCP / TCC counters per dispatch of several kernels from ONE set of five --pmc passes over bench.py (kernel-trace only, separate passes)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
out=$1; shift
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES"
P3="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"
P4="TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM"
P5="GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1)); rm -rf /tmp/pbm$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pbm$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-parity-leg --no-fp32-grad-leg --repeats 1 "${args[@]}" > /dev/null 2>/tmp/pbm$i.err || { echo "pass $i failed"; tail -3 /tmp/pbm$i.err; }
done
for spec in "$@"; do
  name=${spec%%=*}; pat=${spec#*=}
  python3 - "$pat" > $out/pmc_$name.txt <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
print("# kernel name contains:", pat)
for i in range(1, 6):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/pbm%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""): continue
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(f"pass{i} {k:36s} per-dispatch {v / max(n, 1):16.1f}  (n={n})")
PY
done
