#!/bin/bash
# usage: tools/collect_extras.sh <tag>  - the secondary measurements of a round (secondary workloads, fp32 mode,
# in-loop text encoder, ring-kernel phase stamps, launch floor) -> gpurun_out/<tag>_extras/
R=$GRAFT_REPO_ROOT; tag=${1:-r03}; O=$R/gpurun_out/${tag}_extras; mkdir -p $O; cd $R
D=$R/multimodal-emotion-recognition_amd/csrc
python3 bench.py --workload c3l24 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3l24_bf16.json 2>> $O/err.txt
python3 bench.py --workload c3 --ragged --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_bf16_ragged.json 2>> $O/err.txt
python3 bench.py --workload c2 --ragged --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c2_bf16_ragged.json 2>> $O/err.txt
python3 bench.py --workload c2 --dtype fp32 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c2_fp32.json 2>> $O/err.txt
python3 bench.py --workload c3 --dtype fp32 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_fp32.json 2>> $O/err.txt
for m in base large; do
  python3 tools/bench_text_encoder.py --model $m > $O/f4_text_encoder_${m}_bf16.json 2>> $O/err.txt
  python3 tools/bench_text_encoder.py --model $m --dtype fp8 > $O/f4_text_encoder_${m}_fp8.json 2>> $O/err.txt
done
python3 tools/bench_text_encoder.py --model large --utterances 1024 --with-fusion-step > $O/f4_c5_dataflow_large_bf16.json 2>> $O/err.txt
python3 bench.py --workload c3 --packed --ragged --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_bf16_ragged_packed.json 2>> $O/err.txt
python3 bench.py --workload c3b256 --secondary none --no-cpu-baseline --no-parity-leg > $O/bench_c3_b256_bf16.json 2>> $O/err.txt
if [ -f $D/libm2fnet_hip_ttiming.so ]; then
  M2F_LIB=$D/libm2fnet_hip_ttiming.so python3 tools/table_timing.py > $O/table_kernel_phase_totals_c3.txt 2>> $O/err.txt
fi
if [ -f $D/libm2fnet_hip_timing.so ]; then
  for sh in "1024 3072 1024 0" "1024 1024 2048 0"; do
    echo "== ring 128x128, shape $sh (stamps in units of 100 cycles)" >> $O/ring_kernel_phase_stamps.txt
    M2F_LIB=$D/libm2fnet_hip_timing.so M2F_TIMING_RING=1 M2F_RING_MIN=1 python3 tools/gemm_timing.py $sh >> $O/ring_kernel_phase_stamps.txt 2>> $O/err.txt
  done
fi
(cd tools/launchfloor && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 launchfloor.hip -o /tmp/launchfloor 2>> $O/err.txt && /tmp/launchfloor > $O/launch_floor.txt 2>> $O/err.txt)
echo done > $O/DONE
