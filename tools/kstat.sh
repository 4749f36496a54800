#!/bin/bash
# usage: tools/kstat.sh <tag> <gemm_one args>  -> prints avg kernel duration from rocprofv3 kernel trace
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp; rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/ks_$tag/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Name"]: print("$tag", "$*", "avg_us", float(r["AverageNs"])/1e3, "calls", r["Calls"])
PY
