"""Accumulated s_memtime phase totals (cycles) of workgroup 0 of the weight-gradient table launch of ONE bf16 training step
(workload = M2F_WORKLOAD, default c3).  Needs the diagnostic build: `make -C multimodal-emotion-recognition_amd/csrc ttiming`
and M2F_LIB=<...>/libm2fnet_hip_ttiming.so.  Consumer role (wave 0): descriptor fetch / k-loops / epilogues summed over the
tiles the workgroup walked; producer role (wave 4): issuing loads / waiting for a k-tile to land / waiting at the barrier
(= for the consumers), summed over all k-tiles."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402

wl = bench.WORKLOADS[os.environ.get("M2F_WORKLOAD", "c3")]
cfg, B, L = wl["cfg"], wl["B"], wl["L"]
torch.manual_seed(0)
m = M2FNet(cfg, precision="bf16").cuda().train()
text, audio, key_pad, emotion = bench.synthetic_batch(cfg, B, L, 0, torch.device("cuda"), False)
plan = m.engine().plan(B, L, True, True)
plan.set_inputs(text, audio, key_pad, emotion)
fn = runtime.lib().m2f_ring_table_dbg_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
for _ in range(3):
    plan.step(0.1, False, False, False)
torch.cuda.synchronize()
assert fn(buf, 1) == 0
plan.step(0.1, False, False, False)
torch.cuda.synchronize()
assert fn(buf, 0) == 0
c = list(buf)
tiles, ktiles = max(c[11], 1), max(c[16 + 11], 1)
print(f"workgroup 0: {tiles} tiles, {ktiles} k-tiles (cycles; per tile / per k-tile in brackets)")
print(f"consumer: descriptors {c[8]} [{c[8] / tiles:.0f}] | k-loops {c[9]} [{c[9] / tiles:.0f} per tile, {c[9] / ktiles:.0f} per k-tile] | "
      f"epilogues {c[10]} [{c[10] / tiles:.0f}] | sum {c[8] + c[9] + c[10]}")
print(f"producer: issuing {c[24]} [{c[24] / ktiles:.0f}] | waiting for data {c[25]} [{c[25] / ktiles:.0f}] | at the barrier {c[26]} "
      f"[{c[26] / ktiles:.0f}] | sum {c[24] + c[25] + c[26]}")
