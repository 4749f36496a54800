#!/bin/bash
# round-4 A/B batch: the eight-phase 256x256 form (gemm_p8.h) against the round-3 forms, one box, one call
set -o pipefail
O=gpurun_out/${1:-r04e}; mkdir -p $O
timeout -k 10 120 python tools/p8_bench.py check > $O/check.log 2>&1; echo "check rc=$?"; tail -1 $O/check.log
timeout -k 10 200 python tools/p8_bench.py race 16 > $O/race.log 2>&1; echo "race rc=$?"; tail -1 $O/race.log
for SK in 0 3000 1073744824; do
  timeout -k 10 200 env M2F_P8_SKEW=$SK python tools/p8_bench.py bench 20 2>&1 | grep -v amdgpu.ids > $O/p8_bench_skew_$SK.log; echo "p8 bench skew=$SK rc=$?"
  python - <<PY
import json
for l in open("$O/p8_bench_skew_$SK.log"):
    d=json.loads(l); print("  ", d["form"][:2], d["M"], d["N"], d["K"], round(d["us"],1), "us", round(d["TFLOP/s"]), "TF")
PY
done
( export M2F_LIB=$PWD/multimodal-emotion-recognition_amd/csrc/libm2fnet_hip_p8timing.so; for S in "1 8192 8192 1024" "0 8192 8192 1024"; do timeout -k 10 120 python tools/p8_timing.py $S 2>&1 | grep -v amdgpu.ids >> $O/p8_phase_totals.txt; done ); cat $O/p8_phase_totals.txt
for T in 131 132 131 132; do
  timeout -k 10 200 env M2F_TABLE_TILE=$T python bench.py --steps 50 --warmup 10 --secondary none --no-parity-leg --no-cpu-baseline --repeats 3 --dump-launches $O/launches_$T.txt > $O/bench_$T.json 2> $O/bench_$T.err; echo "bench($T) rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$T.json"))
print("tile $T: ms/step", [round(x,3) for x in d["repeats"]["ms_per_step"]], "fwd_bwd", round(d["fwd_bwd_only"]["ms_per_step"],3), "gemm frac", round(d["roofline"]["frac"],4))
PY
done
grep gemm_wgrad $O/launches_131.txt | tail -1; grep gemm_wgrad $O/launches_132.txt | tail -1
for P in 1 0; do
  timeout -k 10 200 env M2F_P8=$P python tools/bench_text_encoder.py --model large --dtype bf16 --steps 10 > $O/text_large_bf16_p8_$P.json 2> $O/text_$P.err; echo "text encoder (M2F_P8=$P) rc=$?"; python -c "import json; d=json.load(open('$O/text_large_bf16_p8_$P.json')); print(d['ms_per_forward'], d['achieved_tflops'])"
done
