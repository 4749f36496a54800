#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
for F in "" "--grad-bf16" "" "--grad-bf16"; do
  N=$([ -z "$F" ] && echo fp32grad || echo bf16grad)
  timeout -k 10 200 python bench.py --steps 50 --warmup 10 --secondary none --no-parity-leg --no-cpu-baseline --repeats 3 $F > $O/bench_$N.json 2> $O/bench_$N.err; echo "bench($N) rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$N.json"))
print("$N: ms/step", [round(x,3) for x in d["repeats"]["ms_per_step"]], "fwd_bwd", round(d["fwd_bwd_only"]["ms_per_step"],3), "value", round(d["value"]), "loss", d["loss"])
PY
done
