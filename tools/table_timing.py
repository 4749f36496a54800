"""Phase timestamps (s_memtime ticks / 100: the counter runs at the ~2.1 GHz shader clock, so printed value x 100 / 2100 = us) of workgroup 0 in the LAST tile it walks in the weight-gradient table kernel of one
C2 bf16 training step; needs an M2F_EXP_TIMING build via M2F_LIB (see tools/README.md)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402

wl = bench.WORKLOADS["c2"]
cfg, B, L = wl["cfg"], wl["B"], wl["L"]
torch.manual_seed(0)
m = M2FNet(cfg, precision="bf16").cuda().train()
text, audio, key_pad, emotion = bench.synthetic_batch(cfg, B, L, 0, torch.device("cuda"), False)
plan = m.engine().plan(B, L, True, True)
plan.set_inputs(text, audio, key_pad, emotion)
for _ in range(3):
    plan.step(0.1, False, False, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn = runtime.lib().m2f_dbg_read
fn.restype = ctypes.c_int
assert fn(buf) == 0
c = list(buf)
t0 = min(c[0], c[16])
print("workgroup 0, kernel start = 0; stamps 1.. are of the LAST tile it walked (ticks / 100)")
print("consumer: start %.2f | at B0 %.2f | B0 passed %.2f | k-loop done %.2f | epilogue done %.2f" % tuple((c[i] - t0) / 100.0 for i in range(5)))
print("producer: start %.2f | setup done %.2f | D stages issued %.2f | stage0 stored %.2f | B0+issue %.2f | k-loop done %.2f" % tuple((c[16 + i] - t0) / 100.0 for i in range(6)))
