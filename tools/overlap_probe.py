"""Do an HBM-bound kernel (fused Adam over N parameters) and an L2 / MFMA-bound kernel (a chip-filling bf16 ring GEMM) overlap when
they run on two streams?  Prints each alone, both back to back on one stream, and both on two streams."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import mer_amd
from mer_amd import functional as F, runtime
n = 104 * 1024 * 1024
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3; m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
M, N, K = 8192, 8192, 1024
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
a16, b16 = F._shadow16(a), F._shadow16(b)
out = torch.empty(M, N, device="cuda")
def gemm(reps=3):
    for _ in range(reps):
        F.gemm(a, b, F.NT, runtime.BF16, out=out, shadows=(a16, None, b16, None), tile=0)
def adam():
    runtime.adam_step(p, g, m, v, 1, 1e-4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
def both_seq():
    gemm(); adam()
def both_par():
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s1):
        s1.wait_event(e); gemm()
    with torch.cuda.stream(s2):
        s2.wait_event(e); adam()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
print("gemm x3 alone  %.1f us (%.0f TFLOP/s)" % (timed(gemm), 3 * 2.0 * M * N * K / timed(gemm) / 1e6))
print("adam alone     %.1f us" % timed(adam))
print("one stream     %.1f us" % timed(both_seq))
print("two streams    %.1f us" % timed(both_par))
