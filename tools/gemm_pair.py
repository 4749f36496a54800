"""Times each C2 GEMM shape with hipEvents around a batch of launches issued through a PRE-BUILT argument list
(host overhead excluded by launching N times back to back and dividing), fp32-source vs bf16-source staging."""
import sys, os, ctypes
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch, mer_amd
from mer_amd import functional as F, runtime
SHAPES = [("qkv   NT", F.NT, 512, 2304, 768), ("oproj NT", F.NT, 512, 768, 768), ("ffn1  NT", F.NT, 512, 2048, 768),
          ("ffn2  NT", F.NT, 512, 768, 2048), ("dgrad NN", F.NN, 512, 768, 768), ("dffn1 NN", F.NN, 512, 768, 2048),
          ("dffn2 NN", F.NN, 512, 2048, 768), ("wgrad TN", F.TN, 768, 768, 512), ("wffn  TN", F.TN, 2048, 768, 512),
          ("wqkv  TN", F.TN, 2304, 768, 512)]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, lay, M, N, K in SHAPES:
    if lay == F.NT: a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    elif lay == F.NN: a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    else: a, b = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    res = {}
    for src16 in (False, True):
        for _ in range(3): F.gemm(a, b, lay, runtime.BF16, out=out, tile=tile, src16=src16)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            a16 = F._shadow16(a); b16 = F._shadow16(b)
            def call():
                runtime.check(runtime.lib().m2f_gemm(runtime.BF16, lay, M, N, K, 0, a.data_ptr(), a.stride(0), None, 0, b.data_ptr(), b.stride(0), None, 0,
                              out.data_ptr(), out.stride(0), None, None, 0, None, 0, 1.0, None, 0, 0, 0, 0, 0, 0.0, None, tile, None, None, 0,
                              a16.data_ptr() if src16 else None, a16.stride(0), None, 0, b16.data_ptr() if src16 else None, b16.stride(0), None, 0,
                              torch.cuda.current_stream().cuda_stream), "gemm")
            call(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(50): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); torch.cuda.synchronize()
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            res[src16] = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * M * N * K
    print(f"tile{tile} {name} M{M} N{N} K{K}: fp32-src {res[False]:7.1f} us ({fl/res[False]/1e6:6.1f} TF)   bf16-src {res[True]:7.1f} us ({fl/res[True]/1e6:6.1f} TF)")
