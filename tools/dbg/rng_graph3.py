import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, bench, mer_amd
from mer_amd.model import M2FNet
wl = bench.WORKLOADS["c2"]
cfg, B, L = dict(wl["cfg"], dropout=0.4), wl["B"], wl["L"]
sd = synth.make_state_dict(cfg)
batch = list(bench.synthetic_batch(cfg, B, L, 0, "cuda:0", ragged=True))
for variant in ("plain", "status", "clone_grad", "clone_loss"):
    os.environ["M2F_MEGA"] = "1"
    torch.manual_seed(11)
    m = M2FNet(cfg, precision="bf16", shape_buckets=False); m.load_state_dict(sd); m = m.to("cuda:0").train()
    out = []
    for step in range(4):
        loss = m.train_step(*batch, use_graph=step > 0)
        torch.cuda.synchronize()
        plan = next(iter(m.engine().plans.values()))
        if variant == "status": plan.check_status()
        if variant == "clone_grad": g = m.engine().flat_grad.clone()
        if variant == "clone_loss": l = plan.loss.clone()
        out.append(round(float(plan.loss[0]), 6))
    print(variant, out)
