#!/usr/bin/env python
"""Where do the persistent kernels (csrc/mega.hip) spend their time?  Needs the diagnostic library:
   make -C multimodal-emotion-recognition_amd/csrc prof
   M2F_LIB=.../libm2fnet_hip_prof.so M2F_MEGA=1 python tools/mega_prof.py --workload c2
Prints, per run (forward / backward chain) and item kind: items, mean item time, mean time in front of the first barrier
(dependency wait + first tile), k-loop, epilogue, and the poll time of the polling lane."""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mer_amd  # noqa
from mer_amd import runtime
from mer_amd.model import M2FNet

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
cfg, B, L = wl["cfg"], wl["B"], wl["L"]
torch.manual_seed(0)
model = M2FNet(cfg, precision="bf16").to("cuda:0").train()
batch = bench.synthetic_batch(cfg, B, L, 0, "cuda:0")
for _ in range(3):
    model.train_step(*batch, use_graph=False)
torch.cuda.synchronize()
plan = next(iter(model.engine().plans.values()))
assert plan.persistent() == 3, "persistent kernels are off (M2F_MEGA=1?)"
lib = runtime.lib()
lib.m2f_plan_prof.restype = ctypes.c_int
lib.m2f_plan_prof.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
out = (ctypes.c_uint64 * 128)()
lib.m2f_plan_prof(plan.handle, out)            # clear
for _ in range(args.steps):
    model.train_step(*batch, use_graph=False)
torch.cuda.synchronize()
plan.check_status()
runtime.check(lib.m2f_plan_prof(plan.handle, out), "m2f_plan_prof")
names = ["(workgroup)", "gemm", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "dropout"]
us = lambda ticks: ticks * 0.01
for run, rn in enumerate(("forward", "backward")):
    t = [[out[run * 64 + k * 8 + f] for f in range(8)] for k in range(8)]
    wgs = t[0][0] / args.steps
    print(f"== {rn}: {wgs:.0f} workgroups, mean workgroup lifetime {us(t[0][1]) / max(t[0][0], 1):.1f} us, longest {us(t[0][2]):.1f} us")
    busy = 0.0
    for k in range(1, 7):
        n = t[k][0]
        if not n:
            continue
        busy += us(t[k][1])
        extra = f" poll(lane) {us(t[k][5]) / n:6.2f}  k-tiles/item {t[k][6] / n:4.1f}" if k == 1 else ""
        print(f"  {names[k]:9s} items/step {n / args.steps:8.0f}  item {us(t[k][1]) / n:6.2f} us = wait+first {us(t[k][2]) / n:6.2f} + body {us(t[k][3]) / n:6.2f} + epilogue {us(t[k][4]) / n:5.2f}{extra}   sum/step/WG {us(t[k][1]) / args.steps / max(wgs, 1):7.1f} us")
    print(f"  item time per workgroup and step: {busy / args.steps / max(wgs, 1):.1f} us")
