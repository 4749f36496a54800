"""Packed ("varlen") plans - m2f_plan_create_packed, runtime.Plan(T=...), M2FNet(packed=True) - against the padded plans.

A packed plan holds only the valid utterances of a ragged batch (token rows cu[b] .. cu[b+1]-1 per dialogue) where the padded
plan holds B x L slots.  Every op but the dialogue attention is token-row-wise and the attention of a dialogue only ever sees
its own valid rows, so with dropout off the valid logits are the SAME BITS in both layouts (each output row is one MFMA
accumulation chain over the same operands in the same order), the loss agrees to fp32 rounding, and the parameter gradients to
the summation order of the token reduction (the padded layout adds its pad slots' exact zeros in between).  Padded plans are
pinned to the oracle / the reference's fixtures by tests/test_model_gpu.py and tests/test_bench_geometry_gpu.py.
"""
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _model(cfg, sd, precision, packed):
    from mer_amd.model import M2FNet
    m = M2FNet(cfg, precision=precision, packed=packed)
    m.load_state_dict(sd)
    return m.to("cuda:0").train()


def _ragged_case(name):
    full = synth._cfg
    if name == "tiny":
        return full(48, 64, 64, 4, 4, 4, 1, 1, 1), 6, 16, [16, 3, 9, 1, 12, 5]
    if name == "two_tiles":                                   # L = 24: two 16-row attention tiles, dialogues crossing them
        return full(96, 128, 64, 4, 8, 4, 1, 2, 2), 7, 24, [24, 3, 17, 20, 9, 1, 11]
    if name == "odd_batch":                                   # B = 5 -> bucket of 8 dialogues: three filler dialogues
        return full(64, 64, 64, 4, 4, 4, 2, 2, 1), 5, 16, [7, 2, 16, 5, 1]
    return full(768, 1024, 768, 8, 8, 8, 2, 2, 2), 16, 16, [16, 9, 4, 12, 1, 7, 16, 3, 10, 5, 8, 2, 14, 6, 11, 9]       # c3_slice width


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["tiny", "two_tiles", "odd_batch", "c3_width"])
def test_packed_plan_equals_padded_plan(name, precision):
    cfg, B, L, lengths = _ragged_case(name)
    cfg = dict(cfg, dropout=0.0)
    sd = synth.make_state_dict(cfg)
    text, audio, key_pad, emotion = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn")]
    valid = ~key_pad
    ref, new = _model(cfg, sd, precision, False), _model(cfg, sd, precision, True)
    l0 = ref.train_step(text, audio, key_pad, emotion, use_graph=False)
    l1 = new.train_step(text, audio, key_pad, emotion, use_graph=False)
    torch.cuda.synchronize()
    p0, p1 = next(iter(ref.engine().plans.values())), next(iter(new.engine().plans.values()))
    assert not p0.packed and p1.packed and p1.T < p0.T, (p0.T, p1.T)
    z0, z1 = p0.logits[valid], p1.logits[valid]
    assert torch.equal(z0, z1), f"valid logits differ: max {(z0 - z1).abs().max().item()}"
    assert (p1.logits[~valid] == 0).all()
    assert abs(l0.item() - l1.item()) <= 2e-6 * max(1.0, abs(l0.item())), (l0.item(), l1.item())
    g0, g1 = ref.engine().flat_grad, new.engine().flat_grad
    tol = 1e-5 if precision == "fp32" else 2e-4
    for (p, o, n, _), name_ in zip(ref.engine().items, [k for k, _ in ref.named_parameters()]):
        a, b = g0[o: o + n], g1[o: o + n]
        scale = a.abs().max().item()
        if scale > 0:
            assert (a - b).abs().max().item() <= tol * scale, (name_, (a - b).abs().max().item(), scale)
    # graph replay of the packed plan == its eager step, and a second batch with other lengths re-uses / re-packs correctly
    l2 = new.train_step(text, audio, key_pad, emotion, use_graph=True)
    l3 = new.train_step(text, audio, key_pad, emotion, use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(l2, l3) and abs(l2.item() - l1.item()) <= 1e-6 * max(1.0, abs(l1.item()))


def test_packed_plan_is_reused_for_batches_of_other_lengths():
    """One packed plan (same B, L, T bucket) serves batches whose dialogues have other lengths: nothing of the previous batch
    may leak through the token rows that belong to no dialogue (they are rewritten as zeros every step)."""
    cfg, B, L, _ = _ragged_case("two_tiles")
    cfg = dict(cfg, dropout=0.0)
    sd = synth.make_state_dict(cfg)
    ref, new = _model(cfg, sd, "bf16", False), _model(cfg, sd, "bf16", True)
    seen = set()
    for step, lengths in enumerate([[24, 20, 6, 20, 20, 15, 8], [24, 15, 23, 9, 15, 15, 5], [24, 1, 1, 2, 24, 24, 24], [24, 20, 6, 20, 20, 15, 8]]):
        text, audio, key_pad, emotion = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn", seed=step)]
        l0 = ref.train_step(text, audio, key_pad, emotion, use_graph=step > 0)
        l1 = new.train_step(text, audio, key_pad, emotion, use_graph=step > 0)
        torch.cuda.synchronize()
        eng = new.engine()
        plan = eng.plans[next(reversed(eng.plans))]
        assert plan.packed
        seen.add(plan.T)
        assert abs(l0.item() - l1.item()) <= 2e-6 * max(1.0, abs(l0.item()))
        g0, g1 = ref.engine().flat_grad, eng.flat_grad
        assert (g0 - g1).abs().max().item() <= 2e-4 * g0.abs().max().item(), (step, (g0 - g1).abs().max().item())
    assert len(new.engine().plans) == len(seen) <= 2          # the plans really were re-used


def test_packed_inference_and_autograd_surface():
    """no-grad forward through a packed plan == padded forward at the valid slots; the autograd surface (loss.backward() on
    the returned logits) scatters d logits into the packed buffer."""
    cfg, B, L, lengths = _ragged_case("two_tiles")
    cfg = dict(cfg, dropout=0.0)
    sd = synth.make_state_dict(cfg)
    text, audio, key_pad, emotion = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn")]
    valid = ~key_pad
    ref, new = _model(cfg, sd, "fp32", False), _model(cfg, sd, "fp32", True)
    with torch.no_grad():
        assert torch.equal(ref(text, audio, key_pad)[valid], new(text, audio, key_pad)[valid])
    crit = torch.nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)
    for m in (ref, new):
        m.zero_grad()
        crit(m(text, audio, key_pad).permute(0, 2, 1), emotion).backward()
    torch.cuda.synchronize()
    g0, g1 = ref.engine().flat_grad, new.engine().flat_grad
    assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()
