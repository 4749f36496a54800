"""bf16-mode plans do not write activation copies nobody reads (csrc/plan.hip::mark_unread_fp32; include/m2fnet_hip.h,
m2f_plan_skipped_copies): the fp32 copy of a QKV projection, an attention output, an attention / FFN-hidden gradient, the
bf16 shadow of a result that is only a residual term or a LayerNorm input.  The unwritten buffers are filled with NaNs when the
plan is built, so a reader the plan builder does not know about would poison the loss and every gradient.  Checked here: the step is bit-for-bit the step that writes every
copy (M2F_SKIP_F32=0 when the plan is built), copies ARE skipped in bf16 mode and never in fp32 mode, and the trajectory of a
few optimizer steps (graph replay, dropout on) stays finite and identical."""
import pytest
import torch

import synth
import mer_amd  # noqa: F401
from mer_amd.model import M2FNet
from mer_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu

CASES = ["c1", "c2_slice", "tiny_odd_heads", "tiny_shared_norm", "tiny_audio_only", "tiny_text_only", "tiny_no_fam", "tiny_ragged",
         "c3_slice_l16", "c3_slice_l24"]


def _run(name, monkeypatch, skip, precision="bf16", steps=1, graph=False):
    monkeypatch.setenv("M2F_SKIP_F32", "1" if skip else "0")
    cfg, B, L, lengths, kind = synth.CASES[name]
    m = M2FNet(cfg, precision=precision)
    m.load_state_dict(synth.make_state_dict(cfg))
    m = m.to("cuda:0").train()
    batch = [x.cuda() for x in synth.make_inputs(cfg, B, L, lengths, kind)]
    opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        losses.append(m.train_step(*batch, use_graph=graph).item())
        if steps > 1:
            opt.step()
    plan = next(iter(m.engine().plans.values()))
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    return losses, grads, plan.skipped_copies(), plan.logits.clone()


@pytest.mark.parametrize("name", CASES)
def test_step_is_bit_identical_with_and_without_the_unread_copies(name, monkeypatch):
    l0, g0, n0, lg0 = _run(name, monkeypatch, skip=False)
    l1, g1, n1, lg1 = _run(name, monkeypatch, skip=True)
    assert n0 == 0 and n1 > 0, (n0, n1)
    assert l0 == l1 and all(map(lambda v: v == v, l1))
    assert torch.equal(lg0, lg1)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g0[k], g1[k]), k


def test_fp32_mode_skips_nothing(monkeypatch):
    _, _, n, _ = _run("c2_slice", monkeypatch, skip=True, precision="fp32")
    assert n == 0


@pytest.mark.parametrize("name", ["c2_slice", "tiny_ragged"])
def test_trajectory_under_graph_replay(name, monkeypatch):
    l0, g0, _, _ = _run(name, monkeypatch, skip=False, steps=5, graph=True)
    l1, g1, n1, _ = _run(name, monkeypatch, skip=True, steps=5, graph=True)
    assert n1 > 0 and l0 == l1
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
