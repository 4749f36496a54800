#!/bin/bash
# usage: tools/traffic.sh <tag> [bench.py args...]
# HBM traffic of the grouped-GEMM launches of one bench step, as MI355X_MICROARCH.md's HBM section prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no tracing domains besides --kernel-trace), the gfx950
# half-count correction of FETCH_SIZE, and a calibration of both counters on a kernel of known byte count in the
# same run (the fused Adam kernel streams 4 reads + 3 writes of 4 B per parameter with 16 B per lane).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/traffic_$tag
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$tag/$C -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-parity-leg --no-fp32-grad-leg --repeats 1 "$@" > $R/gpurun_out/traffic_$tag/$C.json 2> $R/gpurun_out/traffic_$tag/$C.err || echo "pass $C failed"
done
python3 - "$R" "$tag" <<'PY'
import csv, glob, collections, json, sys
R, tag = sys.argv[1], sys.argv[2]
bench = json.loads(open(f"{R}/gpurun_out/traffic_{tag}/FETCH_SIZE.json").read().strip().splitlines()[-1])
n_params = bench["config"]["params"]
per = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{R}/gpurun_out/traffic_{tag}/{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != C: continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    per[C] = {k: (v, n) for k, (v, n) in agg.items()}
def avg(C, pred):
    v = sum(x[0] for k, x in per[C].items() if pred(k)); n = sum(x[1] for k, x in per[C].items() if pred(k))
    return (v / n if n else 0.0), n
is_gemm = lambda k: "m2f_gemm" in k or "m2f_mega" in k
is_adam = lambda k: "m2f_adam" in k
adam_rd, n_ad = avg("FETCH_SIZE", is_adam); adam_wr, _ = avg("WRITE_SIZE", is_adam)
# bytes (padding of the flat buffer ignored: <0.1 %).  The shadow-writing optimizer kernel (bf16 mode, N = 1) also writes W and W^T
# as bf16: 2 x 2 B for every parameter of a 2-D tensor - all but 0.1 % of them
shadowed = any("m2f_adam_shadow" in k for k in per["WRITE_SIZE"])
g16 = any("m2f_adam" in k and "<true>" in k for k in per["FETCH_SIZE"])          # bf16 gradients (bench.py's default since round 4): 2 B instead of 4
known_rd, known_wr = (14.0 if g16 else 16.0) * n_params, (16.0 if shadowed else 12.0) * n_params
cal_rd = known_rd / adam_rd if adam_rd else None                # bytes per FETCH_SIZE count incl. the gfx950 1/2 factor
cal_wr = known_wr / adam_wr if adam_wr else None
g_rd, n_g = avg("FETCH_SIZE", is_gemm); g_wr, _ = avg("WRITE_SIZE", is_gemm)
sys.path.insert(0, R)
import bench as bench_py
out = {"source_hash": bench_py.source_hash(), "workload": bench["config"]["workload"], "dtype": bench["dtype"], "gemm_dispatches_counted": n_g,
       "adam_dispatches_counted": n_ad,
       "raw_counter_per_launch": {"gemm_FETCH_SIZE": g_rd, "gemm_WRITE_SIZE": g_wr, "adam_FETCH_SIZE": adam_rd, "adam_WRITE_SIZE": adam_wr},
       "calibration_bytes_per_count": {"FETCH_SIZE": cal_rd, "WRITE_SIZE": cal_wr,
                                       "note": "fused Adam kernel: %d B read + %d B written per parameter; FETCH factor includes the gfx950 x2 correction" % (14 if g16 else 16, 16 if shadowed else 12)},
       "gemm_hbm_bytes_per_launch": {"read": g_rd * (cal_rd or 0), "write": g_wr * (cal_wr or 0)},
       "by_kernel_raw": {C: {k: {"sum": v, "n": n} for k, (v, n) in sorted(per[C].items())} for C in per}}
out["traffic_bytes_per_launch"] = out["gemm_hbm_bytes_per_launch"]["read"] + out["gemm_hbm_bytes_per_launch"]["write"]
out["by_kernel_raw"] = {C: {k: v for k, v in d.items() if "m2f_" in k} for C, d in out["by_kernel_raw"].items()}
json.dump(out, open(f"{R}/gpurun_out/traffic_{tag}/traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("raw_counter_per_launch", "calibration_bytes_per_count", "gemm_hbm_bytes_per_launch", "traffic_bytes_per_launch")}, indent=1))
PY
