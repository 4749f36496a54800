"""debug: the plan-cache test's batch sequence with packed plans on: where do bucketed-packed and exact-padded gradients part?"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")))
import torch, synth
import mer_amd
from mer_amd.model import M2FNet
cfg = synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1)
sd = synth.make_state_dict(cfg)
use_graph = os.environ.get("G", "1") == "1"
m = M2FNet(cfg, packed=True); m.load_state_dict(sd); m = m.cuda().train()
exact = M2FNet(cfg, shape_buckets=False, packed=False); exact.load_state_dict(sd); exact = exact.cuda().train()
g = torch.Generator().manual_seed(0)
for L in list(range(3, 34, 3)) + [33, 16, 17]:
    B = int(torch.randint(3, 9, (1,), generator=g))
    lengths = [int(x) for x in torch.randint(1, L + 1, (B,), generator=g)]
    lengths[0] = L
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn", seed=L)]
    loss = m.train_step(*batch, use_graph=use_graph)
    ref = exact.train_step(*batch, use_graph=False)
    torch.cuda.synchronize()
    eng, eng_x = m.engine(), exact.engine()
    plan = eng.plans[next(reversed(eng.plans))]
    d = (eng.flat_grad - eng_x.flat_grad).abs().max().item(); s = eng_x.flat_grad.abs().max().item()
    print(f"L={L:2d} B={B} lengths={lengths} plan {plan.B}x{plan.L} packed={plan.packed} T={plan.T} nplans={len(eng.plans)} loss {loss.item():.6f} {ref.item():.6f} grad diff {d:.3e} / {s:.3e}", flush=True)
    if d > 1e-5 * s:
        names = [k for k, _ in m.named_parameters()]
        for (prm, o, n, _), name in zip(eng.items, names):
            dd = (eng.flat_grad[o:o+n] - eng_x.flat_grad[o:o+n]).abs().max().item(); ss = eng_x.flat_grad[o:o+n].abs().max().item()
            if dd > 1e-5 * max(ss, 1e-9): print(f"      {name:55s} diff {dd:.3e} scale {ss:.3e}")
        break
