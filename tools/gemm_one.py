"""Run ONE GEMM shape many times (for rocprofv3 PMC passes).  usage: gemm_one.py <bf16|fp32> <NT|NN|TN> M N K [tile] [iters]"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F, runtime
prec = runtime.BF16 if sys.argv[1] == "bf16" else runtime.F32
lay = {"NT": F.NT, "NN": F.NN, "TN": F.TN}[sys.argv[2]]
M, N, K = map(int, sys.argv[3:6])
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 64
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
if lay == F.NT: a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
elif lay == F.NN: a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
else: a, b = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(iters):
    F.gemm(a, b, lay, prec, out=out, tile=tile)
torch.cuda.synchronize()
