#!/bin/bash
# usage: tools/collect_r04.sh <part: 1|2> [tag]   (on the MI355X box) - everything profiles/ holds for round 4 -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; part=${1:-1}; tag=${2:-r04}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
D=$R/multimodal-emotion-recognition_amd/csrc
if [ "$part" = "1" ]; then
  timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
  for w in c3 c2; do
    tools/kstats_bench.sh ${tag}_$w --workload $w --secondary none > $O/kernel_stats_${w}_bf16.txt 2>&1
    python3 tools/kstats_summary.py $(find gpurun_out/ks_${tag}_$w -name p_kernel_stats.csv | head -1) $O/kstats_${w}_bf16.json > /dev/null 2>$O/kstats_${w}.err
    cp $(find gpurun_out/ks_${tag}_$w -name p_kernel_stats.csv | head -1) $O/kernel_stats_${w}_bf16.csv
    cp gpurun_out/ks_${tag}_$w/bench.json $O/bench_under_rocprof_${w}_bf16.json
    tools/traffic.sh ${tag}_$w --workload $w --secondary none > $O/traffic_${w}.txt 2>&1
    cp gpurun_out/traffic_${tag}_$w/traffic.json $O/traffic_${w}_bf16.json
    python3 bench.py --workload $w --secondary none --no-cpu-baseline --no-parity-leg --dump-launches $O/launches_${w}_bf16.txt > $O/bench_${w}_bf16.json 2> $O/bench_${w}.err
    echo "done $w"
  done
  tools/pmc_bench_multi.sh $O --workload c3 --secondary none -- "ring_128x64_c3=ring_kernel<128, 64" "ring_128x128_c3=ring_kernel<128, 128, 4, false" "p8_table_rc_c3=m2f_gemm_p8_kernel<true, true, 1" "attn_bwd_c3=m2f_attn_bwd" "ln_bwd_c3=m2f_ln_bwd" "adam_c3=m2f_adam_shadow"
  echo done > $O/DONE1
else
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
  python3 bench.py --workload c3l24 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3l24_bf16.json 2>> $O/err.txt
  python3 bench.py --workload c3 --ragged --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_bf16_ragged.json 2>> $O/err.txt
  python3 bench.py --workload c3 --packed --ragged --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_bf16_ragged_packed.json 2>> $O/err.txt
  python3 bench.py --workload c2 --dtype fp32 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c2_fp32.json 2>> $O/err.txt
  python3 bench.py --workload c3 --dtype fp32 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_fp32.json 2>> $O/err.txt
  python3 bench.py --workload c3b256 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_b256_bf16.json 2>> $O/err.txt
  python3 bench.py --workload c3 --grad-fp32 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_bf16_grad_fp32.json 2>> $O/err.txt
  for m in base large; do
    python3 tools/bench_text_encoder.py --model $m > $O/f4_text_encoder_${m}_bf16.json 2>> $O/err.txt
    python3 tools/bench_text_encoder.py --model $m --dtype fp8 > $O/f4_text_encoder_${m}_fp8.json 2>> $O/err.txt
  done
  M2F_ROBERTA_FAT=1 python3 tools/bench_text_encoder.py --model large > $O/f4_text_encoder_large_bf16_round3_dataflow.json 2>> $O/err.txt
  M2F_ROBERTA_FAT=1 python3 tools/bench_text_encoder.py --model large --dtype fp8 > $O/f4_text_encoder_large_fp8_round3_dataflow.json 2>> $O/err.txt
  for m in bf16 fp8; do tools/kstats_cmd.sh ${tag}_f4_$m tools/bench_text_encoder.py --model large --dtype $m > $O/f4_text_encoder_large_kernel_stats_$m.txt 2>&1; done
  python3 tools/bench_text_encoder.py --model large --utterances 1024 --with-fusion-step > $O/f4_c5_dataflow_large_bf16.json 2>> $O/err.txt
  python3 tools/bench_text_encoder.py --model large --utterances 1024 --with-fusion-step --dtype fp8 > $O/f4_c5_dataflow_large_fp8.json 2>> $O/err.txt
  python3 tools/ln_stats_ab.py 2>&1 | grep -v amdgpu.ids > $O/ln_stats_ab.txt
  python3 tools/p8_bench.py bench 20 2>&1 | grep -v amdgpu.ids > $O/p8_kernel_bench.txt
  tools/store_probe/store_probe 4096 > $O/store_shape_probe.txt 2>&1
  python3 tools/chain_floor.py c3 2>&1 | grep -v amdgpu.ids > $O/chain_gemm_floor_c3.txt
  if [ -f $D/libm2fnet_hip_p8timing.so ]; then
    rm -f $O/p8_phase_totals.txt
    for S in "1 8192 8192 1024" "0 32768 4096 1024" "0 32768 1024 1024 res" "0 32768 1024 4096 res" "1 4096 4096 4096"; do M2F_LIB=$D/libm2fnet_hip_p8timing.so python3 tools/p8_timing.py $S 2>&1 | grep -v amdgpu.ids >> $O/p8_phase_totals.txt; done
  fi
  echo done > $O/DONE2
fi
