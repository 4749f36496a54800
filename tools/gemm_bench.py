"""Micro-benchmark of the grouped GEMM kernel through the C ABI (single problems, the shapes of the C2 step)."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F, runtime

SHAPES = [  # (name, layout, M, N, K)
    ("qkv   NT", F.NT, 512, 2304, 768), ("oproj NT", F.NT, 512, 768, 768), ("ffn1  NT", F.NT, 512, 2048, 768),
    ("ffn2  NT", F.NT, 512, 768, 2048), ("cls   NT", F.NT, 512, 7, 768),
    ("dgrad NN", F.NN, 512, 768, 768), ("dffn1 NN", F.NN, 512, 768, 2048), ("dffn2 NN", F.NN, 512, 2048, 768),
    ("wgrad TN", F.TN, 768, 768, 512), ("wffn  TN", F.TN, 2048, 768, 512), ("wqkv  TN", F.TN, 2304, 768, 512),
]

def run(prec, tile, iters=200, src16=False):
    KW = {}
    for name, lay, M, N, K in SHAPES:
        if lay == F.NT: a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        elif lay == F.NN: a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        else: a, b = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        flush = torch.empty(64 * 1024 * 1024, device="cuda")   # 256 MB: evict L2/MALL between timed launches
        for _ in range(3): F.gemm(a, b, lay, prec, out=out, tile=tile, **KW)
        torch.cuda.synchronize()
        # warm (back-to-back) timing
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): F.gemm(a, b, lay, prec, out=out, tile=tile, **KW)
        e1.record(); torch.cuda.synchronize()
        warm = e0.elapsed_time(e1) / iters * 1e3
        # cold timing: flush caches before each launch
        tot = 0.0
        for _ in range(20):
            flush.zero_()
            e0.record(); F.gemm(a, b, lay, prec, out=out, tile=tile, **KW); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        cold = tot / 20 * 1e3
        fl = 2.0 * M * N * K
        print(f"{'bf16' if prec else 'fp32'} tile{tile:3d} {name} M{M} N{N} K{K}: warm {warm:7.1f} us ({fl/warm/1e6:7.1f} TF)  cold {cold:7.1f} us")

if __name__ == "__main__":
    precs = (runtime.BF16, runtime.F32) if len(sys.argv) < 2 else ((runtime.BF16,) if sys.argv[1] == "bf16" else (runtime.F32,))
    tiles = (64, 128) if len(sys.argv) < 3 else (int(sys.argv[2]),)
    for prec in precs:
        for tile in tiles:
            run(prec, tile)
