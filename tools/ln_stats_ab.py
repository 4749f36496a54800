#!/usr/bin/env python
"""Would a LayerNorm forward get faster if its row statistics came out of the preceding GEMM's epilogue?  (round-3 review item 4)

The C3 step's merged LayerNorm-forward launch (1,024 token rows; text d = 1,024 + audio d = 768 in ONE launch, as the plans merge them), 200 dependent launches
captured in one hipGraph (launch i + 1 normalises what launch i wrote - the chain the step has), replayed; per launch:
  A  the shipped kernel: row in registers, two wave reductions (mean, variance), normalise, store fp32 + bf16
  B  the same kernel reading (mean, rstd) from memory instead of computing them (m2f_layernorm_fwd_diag(pre = 1)): no reduction at all
and, for scale, an EMPTY-ish kernel chain (the rng-advance kernel: one thread) = what a kernel boundary inside a graph costs here."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402

runtime.require_gpu()
dev = torch.device("cuda:0")
lib = runtime.lib()
T, dims, N = 1024, [1024, 768], 200
x = [torch.randn(T, d, device=dev) for d in dims]
y = [torch.empty(T, d, device=dev) for d in dims]
gam = [torch.rand(d, device=dev) + 0.5 for d in dims]
bet = [torch.randn(d, device=dev) for d in dims]
st = [torch.zeros(T, 2, device=dev) for _ in dims]
ws = torch.zeros(1, device=dev)


def arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


d_arr = (ctypes.c_int * 2)(*dims)


def launch(src, dst, pre):
    runtime.check(lib.m2f_layernorm_fwd_diag(T, 2, d_arr, arr(src), arr(gam), arr(bet), arr(dst), arr(st), 1e-5, pre, runtime.stream_ptr()), "m2f_layernorm_fwd_diag")


def chain(pre):
    for i in range(N):
        launch(x if i % 2 == 0 else y, y if i % 2 == 0 else x, pre)


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()                                             # eager warm-up on this stream
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N * 1e6)
    return best


rng = torch.zeros(4, dtype=torch.int32, device=dev)


def empty_chain():
    for _ in range(N):
        runtime.check(lib.m2f_rng_advance(rng.data_ptr(), runtime.stream_ptr()), "m2f_rng_advance")


launch(x, y, 0)                                          # statistics of x for the pre = 1 runs (values do not matter for the timing)
torch.cuda.synchronize()
out = {"rows": T, "d": dims, "launches_per_graph": N}
out["A_statistics_computed_us_per_launch"] = graph_time(lambda: chain(0))
out["B_statistics_read_us_per_launch"] = graph_time(lambda: chain(1))
out["A_again"] = graph_time(lambda: chain(0))
try:
    out["one_thread_kernel_us_per_launch"] = graph_time(empty_chain)
except Exception as e:                                   # noqa: BLE001
    out["one_thread_kernel_us_per_launch"] = str(e)
bytes_per_launch = sum(T * d * (4 + 4 + 2) for d in dims)
out["bytes_per_launch"] = bytes_per_launch
out["byte_floor_us_at_5TBps"] = bytes_per_launch / 5e12 * 1e6
print(json.dumps(out, indent=1))
