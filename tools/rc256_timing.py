"""Phase totals (cycles) of waves 1 and 5 of workgroup 0 of the 256x256 weight-gradient table launch (M2F_TABLE_TILE=132) of one C3
step; needs a library whose gemm_rc256.hip was compiled with -DM2F_EXP_TIMING (M2F_LIB)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ["M2F_TABLE_TILE"] = "132"
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
fn = runtime.lib().m2f_rc256_dbg_read
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
for name, o in (("wave 1 (loads A)", 0), ("wave 5 (loads B)", 16)):
    kt = max(c[o + 13], 1)
    print(f"{name}: {kt} k-tiles | own loads {c[o + 8] / kt:.0f} | barrier {c[o + 9] / kt:.0f} | issue {c[o + 10] / kt:.0f} | "
          f"fragments + MFMA {c[o + 11] / kt:.0f} (per k-tile) | epilogues {c[o + 12]} total = {c[o + 12] / (kt / 16):.0f} per tile")
