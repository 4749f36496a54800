import os, sys, torch, numpy as np
sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import synth
from mer_amd.model import M2FNet
def run(name, skip):
    os.environ["M2F_SKIP_F32"] = "1" if skip else "0"
    cfg, B, L, lengths, kind = synth.CASES[name]
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="bf16").cuda().train()
    batch = [x.cuda() for x in synth.make_inputs(cfg, B, L, lengths, kind)]
    loss = m.train_step(*batch, use_graph=False)
    return loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}
for name in sys.argv[1:]:
    l0, g0 = run(name, False); l1, g1 = run(name, True)
    bad = [(k, float((g0[k] - g1[k]).abs().max())) for k in g0 if not torch.equal(g0[k], g1[k])]
    print(name, "loss", l0, l1, "mismatching grads:", len(bad), bad[:12])
