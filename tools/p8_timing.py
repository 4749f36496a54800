"""Accumulated s_memtime phase totals (cycles) of workgroup 0 of the eight-phase 256x256 GEMM (csrc/gemm_p8.h) on one shape.
Needs the diagnostic build: `make -C multimodal-emotion-recognition_amd/csrc p8timing` and
M2F_LIB=<...>/libm2fnet_hip_p8timing.so.  usage: python tools/p8_timing.py [rc M N K]   (default 1 8192 8192 1024)
Per phase of a k-tile and per wave half (wave 0 = upper half, wave 4 = lower half, one barrier behind):
  A = phase start .. fragments in registers (issue of the reads + one prefetch half-tile, the opening barrier, the wait for the reads)
  B = the 16 MFMAs' issue   C = the closing barrier.   Stamps cost ~40 cycles each and drain the LDS queue: read SHARES."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import p8_bench as PB  # noqa: E402

rc, M, N, K = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 8192, 8192, 1024)
fn = runtime.lib().m2f_p8_dbg_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
a, b = PB.operands(rc, M, N, K, 3)
c = torch.empty(M, N, device="cuda")
res = torch.randn(M, N, device="cuda") if (len(sys.argv) > 5 and sys.argv[5] == "res" and not rc) else None      # (k-contiguous form: + residual)
for _ in range(3):
    PB.run(rc, M, N, K, a.view(torch.int16), b.view(torch.int16), c, res=res)
torch.cuda.synchronize()
assert fn(buf, 1) == 0
PB.run(rc, M, N, K, a.view(torch.int16), b.view(torch.int16), c, res=res)
torch.cuda.synchronize()
assert fn(buf, 0) == 0
v = list(buf)
print(f"p8 {'RC' if rc else 'KC'} {M}x{N}x{K}{' + residual' if res is not None else ''}: workgroup 0")
for h, name in ((0, "wave 0 (upper half)"), (1, "wave 4 (lower half)")):
    o = h * 16
    kt = max(v[o + 12], 1)
    tot = sum(v[o: o + 12])
    print(f"  {name}: {kt} k-tiles, {tot / kt:.0f} cycles per k-tile in the phases (MFMA minimum 1,024 per wave, 2,048 per SIMD pair), epilogues {v[o + 13]} cycles in all")
    for p in range(4):
        A, B_, C_ = v[o + 3 * p] / kt, v[o + 3 * p + 1] / kt, v[o + 3 * p + 2] / kt
        print(f"    phase {p + 1}: reads+stage+barrier+wait {A:7.0f} | 16 MFMAs {B_:6.0f} | closing barrier {C_:6.0f} | sum {A + B_ + C_:7.0f}")
    nt = max(v[32 + o + 6], 1)
    nk_tile = kt // nt
    b = [v[32 + o + i] / nt for i in range(6)]
    print(f"    by position inside an output tile ({nt} tiles of {nk_tile} k-tiles; stamps included): k-tile 0 {b[0]:.0f}, 1 {b[1]:.0f}, 2 {b[2]:.0f}, 3 {b[3]:.0f}, "
          f"4-7 {b[4] / max(1, min(4, nk_tile - 4)):.0f} each, 8+ {b[5] / max(1, nk_tile - 8):.0f} each; epilogue {v[o + 13] / nt:.0f} per tile")
