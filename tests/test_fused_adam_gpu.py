"""Optimizer inside the step (round 4): in bf16 mode the weight-gradient launch (csrc/gemm_p8.h, EPI 3) applies torch.optim.Adam's update
(src/train.py:56,231 of the reference) to the elements whose gradient it holds in registers and one more launch inside the same graph
updates the rest - dW never reaches memory.  Same arithmetic on the same gradients as the optimizer's own kernel: every comparison here
is BIT FOR BIT against the two-launch path (m2f_step, then m2f_adam_step_shadowed), which stays selectable:
  * parameters, both moments, both bf16 parameter shadows and the losses over several steps (eager, capture, replays), on geometries with
    whole and partial 256 x 256 tiles, a shared final LayerNorm, a model without fusion stack, the 300-wide audio operand;
  * with the gradients divided by a device-side denominator (the data-parallel / bench form of the step: normalise = 0);
  * a learning-rate change between steps reaches the captured graph (the step-dependent factors live in device memory);
  * fp32 models fall back to the optimizer's own kernel without being asked."""
import pytest
import torch

import synth
import mer_amd  # noqa: F401
from mer_amd.model import M2FNet
from mer_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu


def _model(cfg, precision="bf16"):
    m = M2FNet(cfg, precision=precision)
    m.load_state_dict(synth.make_state_dict(cfg))
    return m.to("cuda:0").train()


def _run(name, fused, steps=5, normalise=True, lr_change=False, precision="bf16"):
    cfg, B, L, lengths, kind = synth.CASES[name]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]
    m = _model(cfg, precision)
    opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
    if not normalise:
        den = torch.tensor([float((batch[3] != -1).sum())], device="cuda")
        opt.grad_scale = den
    losses = []
    for i in range(steps):
        if lr_change and i == 3:
            opt.param_groups[0]["lr"] = 2.5e-4
        if fused:
            losses.append(float(m.train_step(*batch, normalise=normalise, use_graph=i > 0, optimizer=opt)))
        else:
            losses.append(float(m.train_step(*batch, normalise=normalise, use_graph=i > 0)))
            opt.step()
    torch.cuda.synchronize()
    eng = m.engine()
    plan = next(iter(eng.plans.values()))
    return {"losses": losses, "p": eng.flat.detach().clone(), "m": opt._m.clone(), "v": opt._v.clone(),
            "sh": eng.wshadow.clone() if eng.wshadow is not None else None, "fresh": eng.shadows_fresh(),
            "armed": getattr(plan, "_fused_key", None) is not None, "err": getattr(plan, "_fused_err", None)}


@pytest.mark.parametrize("name", ["tiny_ragged", "c2_slice", "c3_slice_l16", "tiny_shared_norm", "tiny_no_fam", "tiny_odd_heads"])
def test_fused_optimizer_step_equals_step_then_optimizer(name):
    a, b = _run(name, True), _run(name, False)
    assert a["armed"] and not b["armed"], a["err"]
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    assert a["losses"][-1] < a["losses"][0]
    for k in ("p", "m", "v"):
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
    n_sh = a["sh"].numel() - 32 * 1024                          # (behind the shadows: the optimizer's tensor table)
    assert torch.equal(a["sh"][:n_sh], b["sh"][:n_sh])
    assert a["fresh"] and b["fresh"]


def test_fused_optimizer_with_a_device_side_gradient_denominator():
    a, b = _run("c2_slice", True, normalise=False), _run("c2_slice", False, normalise=False)
    assert a["armed"]
    for k in ("p", "m", "v"):
        assert torch.equal(a[k], b[k]), k
    # (dividing by the denominator inside the optimizer instead of inside the criterion rounds the gradients differently, and Adam's
    #  m / sqrt(v) turns last-bit differences of tiny gradients into visible ones: the two forms of the step only track each other)
    c = _run("c2_slice", False, normalise=True)
    moved = float((c["p"] - _model(synth.CASES["c2_slice"][0]).engine().flat).double().norm())
    assert float((a["p"] - c["p"]).double().norm()) < 0.1 * moved


def test_learning_rate_changes_reach_the_captured_graph():
    a, b = _run("tiny_ragged", True, steps=6, lr_change=True), _run("tiny_ragged", False, steps=6, lr_change=True)
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["m"], b["m"])
    c = _run("tiny_ragged", True, steps=6, lr_change=False)
    assert not torch.equal(a["p"], c["p"])


def test_fp32_models_take_the_optimizers_own_kernel():
    a, b = _run("tiny_ragged", True, precision="fp32"), _run("tiny_ragged", False, precision="fp32")
    assert not a["armed"]
    assert a["losses"] == b["losses"] and torch.equal(a["p"], b["p"])


def test_matrix_gradients_are_not_written_by_the_fused_step():
    """What the docstrings promise: the table's weight gradients never reach memory (the buffer keeps what it held), every other gradient
    (biases, LayerNorm) is written as before."""
    cfg, B, L, lengths, kind = synth.CASES["c2_slice"]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]
    m = _model(cfg)
    opt = FusedAdam(m, lr=1e-3)
    m.train_step(*batch, use_graph=False)                          # fills every gradient
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    m.engine().flat_grad.fill_(7.0)
    m.train_step(*batch, use_graph=False, optimizer=opt)
    torch.cuda.synchronize()
    untouched = written = 0
    for k, p in m.named_parameters():
        g = p.grad
        if g.dim() == 2 and bool((g == 7.0).all()):
            untouched += 1
        else:
            assert torch.equal(g, ref[k]) or g.dim() == 2, k       # (1-D gradients: the same bits as the unfused step's)
            written += 1
    assert untouched >= 10 and written >= 10, (untouched, written)


@pytest.mark.parametrize("name", ["tiny_ragged", "c2_slice", "c3_slice_l16", "tiny_shared_norm"])
def test_gradients_left_as_bf16_equal_the_rounded_fp32_gradients(name):
    """m2f_plan_grad_bf16 (the data-parallel bf16 exchange's buffer filled by the step itself): the weight-gradient launch writes its dW as
    bf16, one cast launch rounds every other gradient - element for element the bits of rounding a plain step's fp32 gradients, with the
    eager and the captured step; switching back restores fp32 gradients."""
    cfg, B, L, lengths, kind = synth.CASES[name]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]
    m = _model(cfg)
    m.train_step(*batch, use_graph=False)
    eng = m.engine()
    plan = next(iter(eng.plans.values()))
    g32 = eng.flat_grad.detach().clone()
    want = g32.to(torch.bfloat16)
    real = torch.zeros(g32.numel(), dtype=torch.bool, device="cuda")
    for (_, o, n, _) in eng.items:
        real[o: o + n] = True
    assert m.set_grad_bf16(True)
    buf16 = eng.grad_bf16_buf
    for use_graph in (False, True, True):
        buf16.zero_()
        eng.flat_grad.fill_(3.0)
        m.train_step(*batch, use_graph=use_graph)
        torch.cuda.synchronize()
        assert torch.equal(buf16[real].view(torch.int16), want[real].view(torch.int16)), use_graph
    n_untouched = sum(1 for (p, o, n, s) in eng.items if len(s) == 2 and bool((eng.flat_grad[o: o + n] == 3.0).all()))
    assert n_untouched >= 4                                        # the table's matrices: no fp32 dW was written
    assert not m.set_grad_bf16(False)
    m.train_step(*batch, use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(eng.flat_grad[real], g32[real])


def test_optimizer_reads_the_bf16_gradients_of_the_step():
    """M2FNet.set_grad_bf16: FusedAdam.step() reads the bf16 buffer the step filled - the parameters after four steps are bit for bit those of
    rounding the fp32 gradient buffer to bf16 between step and optimizer (m2f_adam_step_g16 on a rounded copy), and close to the fp32-gradient
    run; the mode switches off again."""
    cfg, B, L, lengths, kind = synth.CASES["c2_slice"]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]

    def run(mode):
        m = _model(cfg)
        opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
        if mode == "g16":
            assert m.set_grad_bf16(True)
        losses = []
        for i in range(4):
            losses.append(float(m.train_step(*batch, use_graph=i > 0)))
            if mode == "round":
                opt.grads_bf16 = m.engine().flat_grad.to(torch.bfloat16)
            opt.step()
        torch.cuda.synchronize()
        return losses, m.engine().flat.detach().clone(), m
    l_g, p_g, m_g = run("g16")
    l_r, p_r, _ = run("round")
    l_f, p_f, _ = run("fp32")
    assert l_g == l_r and torch.equal(p_g, p_r)
    assert max(abs(a - b) for a, b in zip(l_g, l_f)) < 2e-3 and l_g[-1] < l_g[0]
    assert not m_g.set_grad_bf16(False)
    m32 = _model(cfg, "fp32")
    assert not m32.set_grad_bf16(True)                               # fp32 models keep fp32 gradients
