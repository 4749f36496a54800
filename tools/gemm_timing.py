"""Phase timestamps (s_memtime, 100 MHz) of workgroup 0 of one bf16 NT GEMM; needs an M2F_EXP_TIMING build via M2F_LIB."""
import sys, os, ctypes
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F, runtime
M, N, K, tile = map(int, sys.argv[1:5])
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
a16 = F._shadow16(a); b16 = F._shadow16(b)
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    F.gemm(a, b, F.NT, runtime.BF16, out=out, shadows=(a16, None, b16, None), tile=tile)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
# M2F_TIMING_RING=1: the stamps of the ring kernel (gemm_ring_128x128.hip keeps its own stamp array); pick a shape that takes the
# 128x128 ring form (>= 200 tiles of 128x128, or M2F_RING_MIN=1)
fn = runtime.lib().m2f_ring_dbg_read if os.environ.get("M2F_TIMING_RING") == "1" else runtime.lib().m2f_dbg_read
fn.restype = ctypes.c_int
assert fn(buf) == 0
c = list(buf)
t0 = min(c[0], c[16])
print("consumer: start %.2f | at B0 %.2f | B0 passed %.2f | k-loop done %.2f | epilogue done %.2f | ep: blk0 in regs %.2f | blk0 stored %.2f | blk1 in regs %.2f (x100 cycles)" % tuple((c[i] - t0) / 100.0 for i in range(8)))
print("producer: start %.2f | setup done %.2f | D stages issued %.2f | stage0 stored %.2f | B0+issue %.2f | k-loop done %.2f  (us)" % tuple((c[16 + i] - t0) / 100.0 for i in range(6)))
