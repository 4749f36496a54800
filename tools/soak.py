"""Stability soak: N training steps (hipGraph replay + fused Adam) on one fixed synthetic C2 batch, bf16 mode.  Prints the loss
every 500 steps; fails on a non-finite loss or parameter, or if the loss is not far below its start (memorising one batch of
512 utterances must work).  Also re-runs the same N steps from the same seed and requires the SAME final loss bit for bit
with dropout off (the step is deterministic: no atomics, fixed-order reductions)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
import mer_amd  # noqa: E402,F401
from mer_amd.model import M2FNet  # noqa: E402
from mer_amd.optim import FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "c2"             # c2 | c3
GRAD_BF16 = len(sys.argv) > 3 and sys.argv[3] == "g16"             # gradients left as bf16 by the step (bench.py's default since round 4)


def run(dropout, n):
    wl = bench.WORKLOADS[WORKLOAD]
    cfg = dict(wl["cfg"], dropout=dropout)
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="bf16").cuda().train()
    opt = FusedAdam(m, lr=5e-5, weight_decay=0.01)
    if GRAD_BF16:
        assert m.set_grad_bf16(True)
    batch = bench.synthetic_batch(cfg, wl["B"], wl["L"], 0, torch.device("cuda"), False)
    first = last = None
    for i in range(n):
        loss = m.train_step(*batch)
        opt.step()
        if i % 500 == 0 or i == n - 1:
            last = float(loss)
            first = last if first is None else first
            print(f"dropout {dropout} step {i:5d} loss {last:.6f}", flush=True)
            assert last == last and abs(last) < 1e4, "non-finite loss"
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters()), "non-finite parameter"
    return first, last


f, l = run(0.4, steps)
assert l < 0.75 * f, (f, l)
a = run(0.0, 400)[1]
b = run(0.0, 400)[1]
assert a == b, ("not deterministic", a, b)
# ... and the in-loop text encoder (round 4: eight-phase bf16 / fp8 GEMMs, bf16 attention, e4m3 operand copies): 20 forwards at dispatch scale, every one
# bit-identical to the first
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import numpy as np  # noqa: E402
import synth_roberta as SR  # noqa: E402
from mer_amd.roberta import RobertaEncoder  # noqa: E402
c = SR.cfg(1024, 2, 16, 4096, 400, 80)
ids, mask = SR.make_batch(c, 192, 64, [int(x) for x in np.random.Generator(np.random.Philox(key=5)).integers(8, 65, size=192)], seed=6)
for prec in ("bf16", "fp8"):
    enc = RobertaEncoder(c, precision=prec)
    enc.load_state_dict(SR.make_state_dict(c))
    enc = enc.cuda().eval()
    first = enc(ids.cuda(), mask.cuda())
    assert torch.isfinite(first[mask.bool().cuda()]).all()
    for _ in range(20):
        assert torch.equal(enc(ids.cuda(), mask.cuda()), first), f"text encoder ({prec}) not deterministic"
    print(f"text encoder {prec}: 20 forwards bit-identical")
print("soak ok")
