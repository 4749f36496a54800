import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth, bench, mer_amd
from mer_amd.model import M2FNet
wl = bench.WORKLOADS["c2"]
cfg, B, L = dict(wl["cfg"], dropout=0.4), wl["B"], wl["L"]
sd = synth.make_state_dict(cfg)
batch = list(bench.synthetic_batch(cfg, B, L, 0, "cuda:0", ragged=True))
res = {}
for name, mega, graph in (("list_eager", 0, False), ("list_graph", 0, True), ("mega_eager", 1, False), ("mega_graph", 1, True)):
    os.environ["M2F_MEGA"] = str(mega)
    torch.manual_seed(11)
    m = M2FNet(cfg, precision="bf16", shape_buckets=False); m.load_state_dict(sd); m = m.to("cuda:0").train()
    outs = []
    for step in range(5):
        m.train_step(*batch, use_graph=graph and step > 0)
        torch.cuda.synchronize()
        plan = next(iter(m.engine().plans.values()))
        outs.append((plan.logits.clone(), m.engine().rng.clone().cpu().tolist()))
    res[name] = outs
for step in range(5):
    base = res["list_eager"][step]
    print(step, "rng", [res[n][step][1] for n in res], "logits equal to list_eager:", {n: bool(torch.equal(res[n][step][0], base[0])) for n in res})
print("---- interleaved engines")
eng = {}
for name, mega in (("list_graph", 0), ("mega_graph", 1)):
    torch.manual_seed(11)
    m = M2FNet(cfg, precision="bf16", shape_buckets=False); m.load_state_dict(sd); m = m.to("cuda:0").train()
    eng[name] = (m, mega)
for step in range(5):
    row = {}
    for name, (m, mega) in eng.items():
        os.environ["M2F_MEGA"] = str(mega)
        m.train_step(*batch, use_graph=step > 0)
        torch.cuda.synchronize()
        plan = next(iter(m.engine().plans.values()))
        row[name] = (bool(torch.equal(plan.logits, res["list_eager"][step][0])), m.engine().rng.cpu().tolist(), plan.persistent())
    print(step, row)
