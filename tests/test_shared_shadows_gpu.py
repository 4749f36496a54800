"""bf16 mode keeps ONE buffer of bf16 parameter shadows per model (W and W^T of every 2-D parameter) that the fused optimizer
writes while it updates the parameters (m2f_adam_step_shadowed), so the forward needs no parameter casts.  Checked here: the
fused kernel = the flat Adam kernel + the cast kernels; training with and without the shared buffer follows the same
trajectory; and every other way of changing parameters brings the casts back (no stale shadow is ever multiplied)."""
import math

import numpy as np
import pytest
import torch

import synth
import mer_amd  # noqa: F401
from mer_amd import layout, runtime
from mer_amd.model import M2FNet
from mer_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu


def _model(cfg, sd=None, precision="bf16"):
    m = M2FNet(cfg, precision=precision)
    m.load_state_dict(sd if sd is not None else synth.make_state_dict(cfg))
    return m.to("cuda:0").train()


def _shadow_offsets(c):
    """(spec, soff, soff_t) of every 2-D parameter: the layout of csrc/plan.hip::pm_add."""
    specs, _ = layout.param_specs(c)
    out, run = [], 0
    r64 = lambda n: (n + 63) // 64 * 64
    p8 = lambda n: (n + 7) // 8 * 8
    for sp in specs:
        if sp.alias_of or len(sp.shape) != 2:
            continue
        rows, cols = sp.shape
        soff = run
        run += r64(rows * p8(cols))
        soff_t = run
        run += r64(cols * p8(rows))
        out.append((sp, soff, soff_t))
    return out, run


@pytest.mark.parametrize("name", ["tiny_odd_heads", "c2_slice", "tiny_shared_norm"])
def test_fused_adam_equals_flat_adam_plus_casts(name):
    cfg = synth.CASES[name][0]
    c = layout.M2FConfig.from_model_config(cfg)
    total = runtime.verify_layout(c)
    g = torch.Generator().manual_seed(3)
    p0 = (torch.rand(total, generator=g) - 0.5).cuda()
    gr = (torch.randn(total, generator=g) * 1e-2).cuda()
    m0 = (torch.randn(total, generator=g) * 1e-3).cuda()
    v0 = (torch.rand(total, generator=g) * 1e-4).cuda()
    scale = torch.tensor([3.0], device="cuda")
    real = torch.zeros(total, dtype=torch.bool, device="cuda")          # the flat buffers pad every tensor to 64 floats: pads are zero
    for sp in layout.param_specs(c)[0]:
        if not sp.alias_of:
            real[sp.offset: sp.offset + sp.numel] = True
    for t in (p0, gr, m0, v0):
        t.mul_(real)
    ref = [t.clone() for t in (p0, m0, v0)]
    runtime.adam_step(ref[0], gr, ref[1], ref[2], 4, 1e-3, (0.9, 0.999), 1e-8, 0.01, scale)
    got = [t.clone() for t in (p0, m0, v0)]
    sh = runtime.param_shadow_buffer(c, torch.device("cuda:0"))
    runtime.adam_step_shadowed(c, got[0], gr, got[1], got[2], sh, 4, 1e-3, (0.9, 0.999), 1e-8, 0.01, scale)
    torch.cuda.synchronize()
    for a, b, what in zip(got, ref, "pmv"):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-9), (what, (a - b).abs().max().item())
    mats, n_sh = _shadow_offsets(c)
    assert len(mats) >= 10
    shadow = sh[:n_sh].view(torch.bfloat16)
    for sp, soff, soff_t in mats:
        rows, cols = sp.shape
        ldd, ldt = (cols + 7) // 8 * 8, (rows + 7) // 8 * 8
        w = got[0][sp.offset: sp.offset + rows * cols].view(rows, cols)
        plain = shadow[soff: soff + rows * ldd].view(rows, ldd)
        trans = shadow[soff_t: soff_t + cols * ldt].view(cols, ldt)
        assert torch.equal(plain[:, :cols], w.to(torch.bfloat16)), sp.name
        assert torch.equal(trans[:, :rows], w.t().to(torch.bfloat16)), sp.name
        assert not plain[:, cols:].float().abs().any() and not trans[:, rows:].float().abs().any(), sp.name   # pads stay zero


def test_training_with_shared_shadows_follows_the_cast_path(monkeypatch):
    """8 optimizer steps in bf16 mode: shared shadows written by the optimizer (default) vs per-plan shadows re-cast at every
    forward (M2F_SHARED_SHADOWS=0).  Same arithmetic up to the contraction of the update's multiply-adds."""
    cfg, B, L, lengths, kind = synth.CASES["c2_slice"]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]

    def run():
        m = _model(cfg)
        opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
        losses = []
        for _ in range(8):
            losses.append(float(m.train_step(*batch, use_graph=True)))
            opt.step()
        return m, losses

    m1, l1 = run()
    eng = m1.engine()
    assert eng.wshadow is not None and eng.shadows_fresh()
    plan = next(iter(eng.plans.values()))
    n_fresh = plan.num_launches()["forward"]
    plan.params_fresh(False)
    assert plan.num_launches()["forward"] > n_fresh                  # the parameter casts are back in the launch list
    monkeypatch.setenv("M2F_SHARED_SHADOWS", "0")
    m2, l2 = run()
    assert m2.engine().wshadow is None
    assert max(abs(a - b) for a, b in zip(l1, l2)) < 5e-4, (l1, l2)
    assert l1[-1] < l1[0] - 0.05
    # (the two optimizer kernels contract the update's multiply-adds differently: an ulp in a parameter now and then lands on the
    #  other side of a bf16 rounding boundary of its shadow, and eight steps amplify that - 2.8e-4 measured)
    d = (m1.flat_parameters() - m2.flat_parameters()).double().norm() / m2.flat_parameters().double().norm()
    assert float(d) < 1e-3, float(d)


def test_stale_shadows_are_never_used():
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]
    sd1 = synth.make_state_dict(cfg, seed=7)
    sd2 = synth.make_state_dict(cfg, seed=8)

    def eval_logits(m):
        m.eval()
        with torch.inference_mode():
            out = m(batch[0], batch[1], batch[2]).clone()
        m.train()
        return out

    m = _model(cfg, sd1)
    opt = FusedAdam(m, lr=1e-3)
    m.train_step(*batch, use_graph=False)
    opt.step()
    eng = m.engine()
    assert eng.shadows_fresh()
    # (1) load_state_dict moves the version counters: the next forward re-casts
    m.load_state_dict({k: v.cuda() for k, v in sd2.items()})
    assert not eng.shadows_fresh()
    ref = eval_logits(_model(cfg, sd2))
    got = eval_logits(m)
    assert torch.equal(got, ref)
    assert eng.shadows_fresh()                                         # a forward that ran the casts leaves them current
    assert torch.equal(eval_logits(m), ref)                            # ... and the next one (no casts) computes the same
    # (2) an in-place edit through torch
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(0.5)
    assert not eng.shadows_fresh()
    half = {k: v * 0.5 for k, v in sd2.items()}
    assert torch.equal(eval_logits(m), eval_logits(_model(cfg, half)))
    # (3) a write through the flat buffer: handing the buffer out invalidates, and an in-place write through a handle taken
    # EARLIER moves the flat buffer's version counter, which the freshness token includes
    m.flat_parameters().mul_(2.0)
    assert not eng.shadows_fresh()
    assert torch.equal(eval_logits(m), ref)
    assert eng.shadows_fresh()
    eng.flat.mul_(0.5)                                                 # (no call to flat_parameters(): the counter alone)
    assert not eng.shadows_fresh()
    assert torch.equal(eval_logits(m), eval_logits(_model(cfg, half)))
    eng.flat[: 64].mul_(1.0)                                           # a view of it
    assert not eng.shadows_fresh()
    # (4) a write torch cannot see (.data) needs the explicit call
    assert torch.equal(eval_logits(m), eval_logits(_model(cfg, half))) and eng.shadows_fresh()
    eng.flat.data.mul_(2.0)
    m.invalidate_shadows()
    assert torch.equal(eval_logits(m), ref)
