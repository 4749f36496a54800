#!/bin/bash
# usage: tools/collect_extras.sh <tag>  - the secondary measurements of a round (persistent kernels, secondary workloads, fp32 mode,
# in-loop text encoder, ring-kernel phase stamps, launch floor) -> gpurun_out/<tag>_extras/
R=$GRAFT_REPO_ROOT; tag=${1:-r02}; O=$R/gpurun_out/${tag}_extras; mkdir -p $O; cd $R
D=$R/multimodal-emotion-recognition_amd/csrc
for w in c3 c2; do
  M2F_MEGA=1 python3 bench.py --workload $w --secondary none --no-cpu-baseline > $O/bench_${w}_bf16_persistent_kernels.json 2> $O/err.txt
  [ -f $D/libm2fnet_hip_prof.so ] && M2F_LIB=$D/libm2fnet_hip_prof.so M2F_MEGA=1 python3 tools/mega_prof.py --workload $w > $O/persistent_kernels_item_profile_${w}.txt 2>> $O/err.txt
done
python3 bench.py --workload c3l24 --secondary none --no-cpu-baseline > $O/bench_c3l24_bf16.json 2>> $O/err.txt
python3 bench.py --workload c3 --ragged --secondary none --no-cpu-baseline > $O/bench_c3_bf16_ragged.json 2>> $O/err.txt
python3 bench.py --workload c2 --ragged --secondary none --no-cpu-baseline > $O/bench_c2_bf16_ragged.json 2>> $O/err.txt
python3 bench.py --workload c2 --dtype fp32 --secondary none --no-cpu-baseline > $O/bench_c2_fp32.json 2>> $O/err.txt
python3 bench.py --workload c3 --dtype fp32 --secondary none --no-cpu-baseline > $O/bench_c3_fp32.json 2>> $O/err.txt
for m in base large; do
  python3 tools/bench_text_encoder.py --model $m > $O/f4_text_encoder_${m}_bf16.json 2>> $O/err.txt
  python3 tools/bench_text_encoder.py --model $m --dtype fp8 > $O/f4_text_encoder_${m}_fp8.json 2>> $O/err.txt
done
python3 tools/bench_text_encoder.py --model large --utterances 1024 --with-fusion-step > $O/f4_c5_dataflow_large_bf16.json 2>> $O/err.txt
if [ -f $D/libm2fnet_hip_timing.so ]; then
  for sh in "1024 3072 1024 0" "1024 1024 2048 0"; do
    echo "== ring 128x128, shape $sh (stamps in units of 100 cycles)" >> $O/ring_kernel_phase_stamps.txt
    M2F_LIB=$D/libm2fnet_hip_timing.so M2F_TIMING_RING=1 M2F_RING_MIN=1 python3 tools/gemm_timing.py $sh >> $O/ring_kernel_phase_stamps.txt 2>> $O/err.txt
  done
fi
(cd tools/launchfloor && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 launchfloor.hip -o /tmp/launchfloor 2>> $O/err.txt && /tmp/launchfloor > $O/launch_floor.txt 2>> $O/err.txt)
echo done > $O/DONE
