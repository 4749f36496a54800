#!/bin/bash
# usage: tools/pmc_mfma.sh [bench args]  -> MFMA-busy and wave-cycle counters per kernel of the bench step (one --pmc pass)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmf
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmf -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph "$@" > /dev/null 2>/tmp/pmf.err || { echo "pass failed"; tail -3 /tmp/pmf.err; }
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        if "m2f_" not in k: continue
        a = agg[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("# per dispatch: MFMA busy cycles (summed over SIMDs), GPU-active cycles (summed over 8 XCDs), MFMA instructions;")
print("# mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)")
for k, c in sorted(agg.items()):
    g = lambda n: c[n][0] / max(c[n][1], 1)
    act = g("GRBM_GUI_ACTIVE") / 8.0
    util = g("SQ_VALU_MFMA_BUSY_CYCLES") / max(act * 256 * 4, 1)
    print(f"{k:62s} n={c['GRBM_GUI_ACTIVE'][1]:5d} mfma_busy={g('SQ_VALU_MFMA_BUSY_CYCLES'):14.0f} active_cyc={act:10.0f} mfma_insts={g('SQ_INSTS_MFMA'):10.0f} mfma_util={util:6.3f}")
PY
