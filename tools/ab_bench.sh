#!/bin/bash
# usage: tools/ab_bench.sh <rounds> "<name>|<ENV=VAL ...>|<bench args>" ...
# Runs bench.py once per variant and round (variants interleaved, so box drift hits all of them alike) and prints, per run:
# ms/step, fwd+bwd ms, GEMM TFLOP/s by events, fam_gemm TFLOP/s and the weight-gradient table launch's event interval.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
mkdir -p $R/gpurun_out/ab
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    name=${v%%|*}; rest=${v#*|}; envs=${rest%%|*}; args=${rest#*|}
    out=$R/gpurun_out/ab/${name}_r$r
    env $envs python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity-leg --repeats 1 --secondary none --dump-launches $out.launches $args > $out.json 2> $out.err || { echo "$name r$r FAILED"; tail -5 $out.err; continue; }
    python3 - "$name" "$r" "$out" <<'PY'
import json, sys
name, r, out = sys.argv[1:4]
d = json.loads(open(out + ".json").read().strip().splitlines()[-1])
wg = [l.split() for l in open(out + ".launches") if "gemm_wgrad" in l]
wg_us = max((float(x[3]) for x in wg), default=0.0)
kinds = {}
for l in open(out + ".launches"):
    f = l.split()
    kinds[f[1]] = kinds.get(f[1], 0.0) + float(f[3])
rf = d["roofline"]
print(f"{name:14s} r{r}: {d['ms_per_step']:.3f} ms/step  fwd+bwd {d['fwd_bwd_only']['ms_per_step']:.3f}  gemm {rf['achieved']:.0f} TF  fam {rf['fam_gemm']['achieved']:.0f} TF  wgrad {wg_us:.0f} us | "
      + " ".join(f"{k}={v:.0f}" for k, v in sorted(kinds.items())), flush=True)
PY
  done
done
