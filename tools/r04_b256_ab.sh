O=gpurun_out/r04_b256; mkdir -p $O
run() { name=$1; shift; env "$@" python3 bench.py --workload c3b256 --steps 30 --warmup 5 --secondary none --no-cpu-baseline --no-parity-leg --repeats 1 --dump-launches $O/launches_$name.txt > $O/$name.json 2> $O/$name.err; python3 -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['ms_per_step'],3), 'fwd_bwd', round(d['fwd_bwd_only']['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"; grep gemm_wgrad $O/launches_$name.txt | tail -1; }
run default A=1
run nop8 M2F_P8=0
run t131 M2F_TABLE_TILE=131
run nop8_t131 M2F_P8=0 M2F_TABLE_TILE=131
