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


def run(dropout, n):
    wl = bench.WORKLOADS["c2"]
    cfg = dict(wl["cfg"], dropout=dropout)
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="bf16").cuda().train()
    opt = FusedAdam(m, lr=5e-5, weight_decay=0.01)
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
print("soak ok")
