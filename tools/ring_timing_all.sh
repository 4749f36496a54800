D=$GRAFT_REPO_ROOT/multimodal-emotion-recognition_amd/csrc
for v in timing timing_128x64 timing_64x64; do
 for sh in "1024 768 768 0" "1024 768 2048 0" "1024 3072 1024 0"; do
  echo "== $v shape $sh"
  M2F_LIB=$D/libm2fnet_hip_$v.so M2F_TIMING_RING=1 M2F_RING_MIN=1 python3 tools/gemm_timing.py $sh
 done
done
