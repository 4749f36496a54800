"""One NT bf16-source GEMM shape, weights cycled through a pool (pool=1: cache-warm; pool*N*K*2 B > 512 MB: HBM-cold).
usage: gemm_cold.py M N K pool [iters]   (run under rocprofv3 --kernel-trace --stats; see tools/kcold.sh)"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F, runtime
M, N, K, pool = map(int, sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 200
lay = os.environ.get("LAYOUT", "NT")
tile = int(os.environ.get("TILE", "0"))
if lay == "NT":
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
elif lay == "NN":
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda")
else:
    a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda")
a16 = F._shadow16(a)
b16 = [F._shadow16(b) + 0 for _ in range(pool)]
out = torch.empty(M, N, device="cuda")
torch.cuda.synchronize()
L = {"NT": F.NT, "NN": F.NN, "TN": F.TN}[lay]
for i in range(iters):
    F.gemm(a, b, L, runtime.BF16, out=out, shadows=(a16, None, b16[i % pool], None), tile=tile)
torch.cuda.synchronize()
