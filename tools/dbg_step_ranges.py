import os, sys, torch
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import synth
from mer_amd.model import M2FNet
from mer_amd.optim import FusedAdam
def run(shared, g16):
    os.environ["M2F_SHARED_SHADOWS"] = "1" if shared else "0"
    cfg, B, L, lengths, kind = synth.CASES["c2_slice"]
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="bf16").cuda().train()
    m.load_state_dict({k: v.cuda() for k, v in synth.make_state_dict(cfg).items()})
    opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
    batch = [x.cuda() for x in synth.make_inputs(cfg, B, L, lengths, kind)]
    eng = m.engine()
    out = []
    for i in range(3):
        m.train_step(*batch, use_graph=False)
        n = eng.flat.numel()
        plan = next(iter(eng.plans.values()))
        split = plan.split_offset()
        starts = sorted(o for (_, o, _, _) in eng.items)
        mid = starts[len(starts) // 3]
        ranges = [(split, n), (mid, split), (0, mid)]
        grads = eng.flat_grad.to(torch.bfloat16) if g16 else None
        opt.step_ranges(ranges, grads=grads)
        torch.cuda.synchronize()
        out.append((eng.flat.clone(), opt._m.clone(), opt._v.clone(), eng.shadows_fresh()))
    return out, [(p, o, nn, s) for (p, o, nn, s) in eng.items], [k for k, _ in m.named_parameters()]
for g16 in (False, True):
    a, items, names = run(True, g16); b, _, _ = run(False, g16)
    for i in range(3):
        for what, x, y in (("p", a[i][0], b[i][0]), ("m", a[i][1], b[i][1]), ("v", a[i][2], b[i][2])):
            if not torch.equal(x, y):
                bad = [(names[j], float((x[o:o+nn] - y[o:o+nn]).abs().max())) for j, (p, o, nn, s) in enumerate(items) if not torch.equal(x[o:o+nn], y[o:o+nn])]
                print("g16", g16, "step", i, what, "differs in", len(bad), "tensors", bad[:6])
        print("g16", g16, "step", i, "fresh", a[i][3], b[i][3])
