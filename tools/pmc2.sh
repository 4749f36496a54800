#!/bin/bash
# usage: tools/pmc2.sh "M N K"  -> SQ/LDS counters per GEMM dispatch (separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
shape="$1"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pm$i -o p -- python3 $R/tools/gemm_cold.py $shape 1 100 > /dev/null 2>/tmp/pm$i.err || { echo "pass $i failed"; tail -3 /tmp/pm$i.err; }
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
        print(f"pass{i} {k:32s} per-dispatch {v / max(n, 1):16.1f}  (n={n})")
PY
