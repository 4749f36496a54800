#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pa$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pa$i -o p -- python3 $R/tools/attn_one.py "$@" > /dev/null 2>/tmp/pa$i.err || { echo "pass $i failed"; tail -3 /tmp/pa$i.err; }
done
rm -rf /tmp/pa3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa3 -o p -- python3 $R/tools/attn_one.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for i in range(1, 3):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/pa%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if "m2f_attn" not in row.get("Kernel_Name", ""): continue
            k = ("bwd " if "bwd" in row["Kernel_Name"] else "fwd ") + row["Counter_Name"]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(f"pass{i} {k:36s} per-dispatch {v / max(n, 1):14.1f}")
for f in glob.glob("/tmp/pa3/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m2f_attn" in r["Name"]: print(r["Name"][28:60], "avg_us %.2f min %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
