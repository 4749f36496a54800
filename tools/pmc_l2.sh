#!/bin/bash
# usage: tools/pmc_l2.sh <kernel-name-substring> [ENV=VAL ...] -- [bench args]
# L1->L2 request count / latency and L2 hit / miss / fabric-read counters per dispatch of one kernel of the bench step
# (two separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
pat="$1"; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for e in "${envs[@]}"; do export "$e"; done
P3="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE"
P4="TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ"
i=0
for P in "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pl$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --secondary none --no-parity-leg --repeats 1 "$@" > /dev/null 2>/tmp/pl$i.err || { echo "pass $i failed"; tail -3 /tmp/pl$i.err; }
done
python3 - "$pat" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
for i in (1, 2):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/pl%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""): continue
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(f"pass{i} {k:36s} per-dispatch {v / max(n, 1):16.1f}  (n={n})")
PY
