#!/bin/bash
# usage: tools/r04_p8_tiles.sh  (on the MI355X box, after `make p8timing`) - where the cycles of an output tile go at K = 1,024: k-tiles by position + epilogue
R=$GRAFT_REPO_ROOT; D=$R/multimodal-emotion-recognition_amd/csrc; O=$R/gpurun_out; mkdir -p $O
for S in "0 32768 4096 1024" "0 32768 1024 1024" "0 32768 1024 1024 res" "0 32768 1024 4096 res" "1 8192 8192 1024"; do
  M2F_LIB=$D/libm2fnet_hip_p8timing.so python3 $R/tools/p8_timing.py $S 2>&1 | grep -v amdgpu.ids
done > $O/p8_tiles.txt
cat $O/p8_tiles.txt
