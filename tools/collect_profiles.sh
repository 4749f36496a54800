#!/bin/bash
# usage: tools/collect_profiles.sh <tag>   (on the MI355X box: gpurun -- 'tools/collect_profiles.sh r02')
# One pass over everything profiles/ holds for a round: rocprofv3 kernel stats of the C3 / C2 bench (+ the JSON digests bench.py
# quotes), HBM traffic of the GEMM launches (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), SQ / TCP / TCC counters of the ring
# GEMM kernels inside the C3 step, the per-launch hipEvent table, and the default bench line.  Results: gpurun_out/<tag>/.
R=$GRAFT_REPO_ROOT; tag=${1:-r03}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for w in c3 c2; do
  tools/kstats_bench.sh ${tag}_$w --workload $w --secondary none > $O/kernel_stats_${w}_bf16.txt 2>&1
  python3 tools/kstats_summary.py $(find gpurun_out/ks_${tag}_$w -name p_kernel_stats.csv | head -1) $O/kstats_${w}_bf16.json > /dev/null 2>$O/kstats_${w}.err
  cp $(find gpurun_out/ks_${tag}_$w -name p_kernel_stats.csv | head -1) $O/kernel_stats_${w}_bf16.csv
  cp gpurun_out/ks_${tag}_$w/bench.json $O/bench_under_rocprof_${w}_bf16.json
  tools/traffic.sh ${tag}_$w --workload $w --secondary none > $O/traffic_${w}.txt 2>&1
  cp gpurun_out/traffic_${tag}_$w/traffic.json $O/traffic_${w}_bf16.json
  python3 bench.py --workload $w --secondary none --no-cpu-baseline --no-parity-leg --dump-launches $O/launches_${w}_bf16.txt > $O/bench_${w}_bf16.json 2> $O/bench_${w}.err
done
tools/pmc_bench.sh "ring_kernel<128, 64" --workload c3 --secondary none > $O/pmc_ring_128x64_c3.txt 2>&1
tools/pmc_bench.sh "ring_kernel<128, 128, 4, false" --workload c3 --secondary none > $O/pmc_ring_128x128_c3.txt 2>&1
tools/pmc_bench.sh "ring_kernel<256, 128, 3, true" --workload c3 --secondary none > $O/pmc_ring_table_rc_256x128_c3.txt 2>&1
tools/pmc_bench.sh "m2f_attn_bwd" --workload c3 --secondary none > $O/pmc_attn_bwd_c3.txt 2>&1
tools/pmc_bench.sh "m2f_ln_bwd" --workload c3 --secondary none > $O/pmc_ln_bwd_c3.txt 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/DONE
