#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r04g}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_fused_adam_gpu.py tests/test_gemm_p8_gpu.py -q -x -m gpu > $O/tests_fused.log 2>&1; echo "fused tests rc=$?"; tail -15 $O/tests_fused.log
for F in "" "--no-fused-adam" "" "--no-fused-adam"; do
  N=$([ -z "$F" ] && echo fused || echo unfused)
  timeout -k 10 200 python bench.py --steps 50 --warmup 10 --secondary none --no-parity-leg --no-cpu-baseline --repeats 3 $F > $O/bench_$N.json 2> $O/bench_$N.err; echo "bench($N) rc=$?"; tail -2 $O/bench_$N.err
  python - <<PY
import json
d=json.load(open("$O/bench_$N.json"))
print("$N: ms/step", [round(x,3) for x in d["repeats"]["ms_per_step"]], "fwd_bwd", round(d["fwd_bwd_only"]["ms_per_step"],3), "value", round(d["value"]), d["config"]["step"][:60])
PY
done
