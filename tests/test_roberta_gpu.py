"""GPU parity of the in-loop text encoder: RobertaEncoder on the HIP kernels vs the fixtures written by the
real transformers.RobertaModel and vs the live CPU oracle.  fp32 mode: 1e-4 on hidden states of O(1) magnitude; bf16 mode
(bf16 GEMM operands, fp32 accumulate, fp32 attention / LayerNorm / GELU): 6e-2 absolute, stated here."""
import os

import numpy as np
import pytest
import torch

import synth_roberta as SR
import mer_amd  # noqa: F401
from mer_amd.roberta import RobertaEncoder
from oracle import roberta_oracle as RO

pytestmark = pytest.mark.gpu


def _enc(c, precision):
    m = RobertaEncoder(c, precision=precision)
    m.load_state_dict(SR.make_state_dict(c))
    return m.cuda().eval()


@pytest.mark.parametrize("name", list(SR.CASES))
def test_fp32_matches_transformers_fixture_and_oracle(golden_dir, name):
    c, B, S, lengths = SR.CASES[name]
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    ids, mask = SR.make_batch(c, B, S, lengths)
    m = _enc(c, "fp32")
    hid = m(ids.cuda(), mask.cuda()).cpu()
    assert np.abs(hid[:, 0, :].numpy() - fx["cls"]).max() < 1e-4
    last = np.stack([hid[b, n - 1].numpy() for b, n in enumerate(lengths)])
    assert np.abs(last - fx["hidden_last_valid"]).max() < 1e-4
    ref = RO.forward(SR.make_state_dict(c), c, ids, mask)
    assert (hid - ref)[mask.bool()].abs().max().item() < 1e-4
    cls = m.cls_embeddings(ids.cuda(), mask.cuda()).cpu()
    assert torch.equal(cls, hid[:, 0, :])


@pytest.mark.parametrize("name", ["roberta_tiny", "roberta_three_blocks", "roberta_base_width"])
def test_bf16_within_stated_tolerance(golden_dir, name):
    c, B, S, lengths = SR.CASES[name]
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    ids, mask = SR.make_batch(c, B, S, lengths)
    hid = _enc(c, "bf16")(ids.cuda(), mask.cuda()).cpu()
    err = np.abs(hid[:, 0, :].numpy() - fx["cls"])
    assert err.max() < 6e-2 and err.mean() < 1e-2, (err.max(), err.mean())


@pytest.mark.parametrize("name", list(SR.CASES))
def test_fp8_within_stated_tolerance(golden_dir, name):
    """fp8 mode (BASELINE C5): the four GEMMs of every layer with OCP e4m3 operands (per-tensor weight scales, fixed
    activation scales), everything else fp32.  Naive per-tensor fp8 costs ~0.06 mean absolute error on O(1) hidden states
    (cosine >= 0.995 against the transformers fixture); stated tolerance: mean 0.1, max 0.5, cosine 0.99."""
    c, B, S, lengths = SR.CASES[name]
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    ids, mask = SR.make_batch(c, B, S, lengths)
    cls = _enc(c, "fp8").cls_embeddings(ids.cuda(), mask.cuda()).cpu().numpy()
    err = np.abs(cls - fx["cls"])
    cos = (cls * fx["cls"]).sum(-1) / (np.linalg.norm(cls, axis=-1) * np.linalg.norm(fx["cls"], axis=-1))
    assert err.mean() < 0.1 and err.max() < 0.5 and cos.min() > 0.99, (err.mean(), err.max(), cos.min())


def test_pad_contents_and_batch_composition_do_not_matter():
    """Valid tokens of a sequence depend neither on what sits in its padded slots nor on the other sequences."""
    c, B, S, lengths = SR.CASES["roberta_two_blocks"]
    ids, mask = SR.make_batch(c, B, S, lengths)
    m = _enc(c, "fp32")
    hid = m(ids.cuda(), mask.cuda()).cpu()
    n1 = lengths[1]
    single = m(ids[1:2, :n1].contiguous().cuda(), mask[1:2, :n1].contiguous().cuda()).cpu()
    assert (single[0] - hid[1, :n1]).abs().max().item() < 1e-5
    assert torch.equal(m(ids.cuda(), mask.cuda()).cpu(), hid)          # replay on the cached workspace: bit-identical


def test_text_encoder_feeds_the_fusion_model():
    """BASELINE C5 data flow: token ids -> RobertaEncoder [CLS] rows -> M2FNet text input."""
    import synth
    from mer_amd.model import M2FNet
    c, B, S, lengths = SR.CASES["roberta_tiny"]                      # hidden 64 == d_text of the tiny M2FNet cases
    cfg, Bm, Lm, mlens, kind = synth.CASES["tiny_ragged"]
    n_utt = Bm * Lm
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, c["vocab_size"], (n_utt, 12), generator=g)
    ids[:, 0] = 0
    enc = _enc(c, "fp32")
    text = enc.cls_embeddings(ids.cuda()).view(Bm, Lm, -1)
    _, audio, key_pad, _ = synth.make_inputs(cfg, Bm, Lm, mlens, kind)
    model = M2FNet(cfg).cuda().eval()
    with torch.inference_mode():
        logits = model(text, audio.cuda(), key_pad.cuda())
    assert logits.shape == (Bm, Lm, 7) and torch.isfinite(logits[~key_pad.cuda()]).all()


def test_fp8_text_encoder_cost_in_downstream_logits():
    """What the fp8 text encoder (BASELINE C5) costs where it matters - in the emotion logits.  [CLS] rows of a RoBERTa-large-WIDTH
    encoder (hidden 1024, 16 heads, 2 layers, synthetic weights) in fp32, bf16 and fp8 mode feed the SAME M2FNet (`c3_slice` geometry
    and weights: d_text 1024, fp32 mode, eval): stated and asserted are the largest logit change and the argmax agreement against the
    fp32 encoder.  (The encoder-level tolerance of test_fp8_within_stated_tolerance - mean 0.1 / max 0.5 on O(1) hidden states - says
    nothing about this by itself.)"""
    import synth
    from mer_amd.model import M2FNet
    cfg, Bm, Lm, mlens, kind = synth.CASES["c3_slice_l16"]
    c = SR.cfg(1024, 2, 16, 4096, 300, 40)
    n_utt, S = Bm * Lm, 24
    g = np.random.Generator(np.random.Philox(key=21))
    lengths = [int(x) for x in g.integers(4, S + 1, size=n_utt)]
    ids, mask = SR.make_batch(c, n_utt, S, lengths, seed=22)
    _, audio, key_pad, _ = synth.make_inputs(cfg, Bm, Lm, mlens, kind)
    model = M2FNet(cfg, precision="fp32")
    model.load_state_dict(synth.make_state_dict(cfg))
    model = model.cuda().eval()
    valid = ~key_pad
    logits = {}
    for prec in ("fp32", "bf16", "fp8"):
        cls = _enc(c, prec).cls_embeddings(ids.cuda(), mask.cuda()).view(Bm, Lm, -1).float()
        cls = cls * 0.63 / cls.std()                                   # the scale of the real text embeddings
        with torch.inference_mode():
            logits[prec] = model(cls, audio.cuda(), key_pad.cuda()).cpu()
    ref = logits["fp32"]
    spread = float((ref[valid].max(-1).values - ref[valid].min(-1).values).mean())
    stats = {}
    for prec in ("bf16", "fp8"):
        d = float((logits[prec] - ref)[valid].abs().max())
        agree = float((logits[prec].argmax(-1) == ref.argmax(-1))[valid].float().mean())
        stats[prec] = (d, agree)
    print(f"downstream logits vs the fp32 encoder (spread of a row's logits {spread:.3f}): bf16 max |d| {stats['bf16'][0]:.4f}, argmax agreement {stats['bf16'][1]:.3f}; fp8 {stats['fp8'][0]:.4f}, {stats['fp8'][1]:.3f}")
    assert stats["bf16"][0] < FP8_LOGIT_BOUNDS["bf16"] and stats["bf16"][1] >= 0.97, (stats, spread)
    assert stats["fp8"][0] < FP8_LOGIT_BOUNDS["fp8"] and stats["fp8"][1] >= 0.90, (stats, spread)
    assert stats["fp8"][0] >= stats["bf16"][0] * 0.5, stats               # fp8 is the coarser mode: the numbers must say so


# measured on MI355X (round 4): a row's logits spread 0.31; bf16 encoder max |dlogit| 2e-4, fp8 encoder 2.8e-3, argmax agreement 1.000 / 1.000 on the 128
# valid utterances.  Bounds: an order of magnitude above the measurement, two below the spread.
FP8_LOGIT_BOUNDS = {"bf16": 2e-3, "fp8": 2e-2}


# ---- round 4: the attention kernel of the encoder's bf16 mode (csrc/attention_long.hip) against an fp32 softmax(QK^T)V of the SAME
# bf16-rounded operands.  What differs: probabilities rounded to bf16 before the P V product (relative 2^-9 each, averaged over the
# keys) and the bf16 rounding of the result itself (2^-9 relative) -> 1.5e-2 of the largest |value| stated, ~4e-3 observed.
ATTN_BF16_TOL = 1.5e-2


@pytest.mark.parametrize("B,S,H,hd", [(3, 64, 4, 64), (2, 100, 2, 32), (2, 130, 3, 16), (1, 200, 2, 128), (2, 37, 2, 24), (2, 64, 1, 8),
                                      (4, 1, 2, 64)])
def test_bf16_attention_kernel_matches_fp32_softmax_of_the_same_operands(B, S, H, hd):
    import ctypes
    from mer_amd import runtime
    g = torch.Generator().manual_seed(S * 131 + hd)
    d = H * hd
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 1.5).cuda().to(torch.bfloat16)      # the packed projection's layout: q | k | v columns
    lengths = [max(1, S - 7 * b) for b in range(B)]
    if B > 2:
        lengths[-1] = 0                                                                  # a sequence with every key padded -> zero rows
    key_pad = torch.ones(B, S, dtype=torch.uint8)
    for b, n in enumerate(lengths):
        key_pad[b, :n] = 0
    key_pad = key_pad.cuda()
    out16 = torch.full((B * S, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    out32 = torch.full((B * S, d), float("nan"), dtype=torch.float32, device="cuda")
    for o32 in (None, out32):
        runtime.check(runtime.lib().m2f_attention_long_fwd_bf16(B, S, H, hd, qkv.data_ptr(), 3 * d, qkv.data_ptr() + 2 * d, 3 * d,
                                                                qkv.data_ptr() + 4 * d, 3 * d, key_pad.data_ptr(), out16.data_ptr(),
                                                                o32.data_ptr() if o32 is not None else None, d, runtime.stream_ptr()),
                      "m2f_attention_long_fwd_bf16")
    torch.cuda.synchronize()
    f = qkv.float().view(B, S, 3, H, hd)
    q, k, v = (f[:, :, i].permute(0, 2, 1, 3) for i in range(3))                         # [B, H, S, hd]
    sc = q @ k.transpose(-1, -2) / hd ** 0.5
    sc = sc.masked_fill(key_pad.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(sc, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)                                                     # (all keys padded)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B * S, d)
    top = ref.abs().max().item()
    assert torch.isfinite(out16.float()).all() and torch.isfinite(out32).all()
    assert (out32 - ref).abs().max().item() < ATTN_BF16_TOL * top, (out32 - ref).abs().max().item() / top
    assert (out16.float() - ref).abs().max().item() < ATTN_BF16_TOL * top, (out16.float() - ref).abs().max().item() / top
    assert torch.equal(out16, out32.to(torch.bfloat16))                                  # one result, two roundings of it
    if B > 2:
        assert out16.view(B, S, d)[-1].float().abs().max().item() == 0.0
    print(f"bf16 attention B={B} S={S} H={H} hd={hd}: max |d| / max |ref| = {(out16.float() - ref).abs().max().item() / top:.2e}")


def test_bf16_mode_lean_and_fat_data_flows_agree():
    """The bf16 mode with bf16-only projections / context / hidden activations (round 4) against the same mode keeping every
    activation as fp32 + bf16 with the fp32-operand attention kernel (rounds 1-3): same weights, same tokens."""
    import mer_amd.roberta as R
    c, B, S, lengths = SR.CASES["roberta_base_width"]
    ids, mask = SR.make_batch(c, B, S, lengths)
    m = _enc(c, "bf16")
    lean = m(ids.cuda(), mask.cuda()).cpu()
    old = R._FAT_BF16
    R._FAT_BF16 = True
    try:
        fat = m(ids.cuda(), mask.cuda()).cpu()
    finally:
        R._FAT_BF16 = old
    ref = RO.forward(SR.make_state_dict(c), c, ids, mask)
    e_lean = (lean - ref)[mask.bool()].abs()
    e_fat = (fat - ref)[mask.bool()].abs()
    print(f"bf16 text encoder vs oracle: lean max {e_lean.max():.3e} mean {e_lean.mean():.3e}; fat max {e_fat.max():.3e} mean {e_fat.mean():.3e}")
    assert e_lean.max().item() < 6e-2 and e_lean.mean().item() < 1e-2
    assert e_lean.mean().item() < 1.5 * e_fat.mean().item() + 1e-4


def test_e4m3_outputs_of_attention_and_layernorm_equal_a_quantise_pass_over_their_fp32_results():
    """fp8 mode (round 4): the attention kernel and the LayerNorm kernel write the e4m3 operand of the next GEMM themselves.  Same bytes
    as m2f_quantize_fp8 over the fp32 result of the same launch (same scale, same saturation)."""
    from mer_amd import runtime
    from mer_amd import functional as F
    B, S, H, hd = 3, 70, 4, 32
    d = H * hd
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 2.0).cuda().to(torch.bfloat16)
    key_pad = torch.zeros(B, S, dtype=torch.uint8)
    key_pad[1, 50:] = 1
    key_pad = key_pad.cuda()
    out32 = torch.full((B * S, d), float("nan"), device="cuda")
    out8 = torch.zeros(B * S, d, dtype=torch.uint8, device="cuda")
    runtime.check(runtime.lib().m2f_attention_long_fwd_bf16_out8(B, S, H, hd, qkv.data_ptr(), 3 * d, qkv.data_ptr() + 2 * d, 3 * d,
                                                                 qkv.data_ptr() + 4 * d, 3 * d, key_pad.data_ptr(), None, out32.data_ptr(),
                                                                 out8.data_ptr(), 16.0, d, runtime.stream_ptr()), "m2f_attention_long_fwd_bf16_out8")
    want = F.quantize_fp8(out32, 16.0)
    torch.cuda.synchronize()
    assert torch.equal(out8, want.view(torch.uint8))
    assert out8.ne(0).float().mean().item() > 0.9                      # (not an all-zero comparison)
    T, dl = 37, 768
    x = (torch.randn(T, dl, generator=g) * 3.0 + 0.5).cuda()
    gam, bet = (torch.rand(dl, generator=g) + 0.5).cuda(), torch.randn(dl, generator=g).cuda()
    gam[:8] *= 40.0                                                     # (some results beyond +-448 / 16: the saturating branch)
    y = torch.empty(T, dl, device="cuda")
    st = torch.empty(T, 2, device="cuda")
    y8 = torch.zeros(T, dl, dtype=torch.uint8, device="cuda")
    runtime.check(runtime.lib().m2f_layernorm_fwd_out8(T, dl, x.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, y.data_ptr(), st.data_ptr(), 1e-5,
                                                       y8.data_ptr(), 16.0, runtime.stream_ptr()), "m2f_layernorm_fwd_out8")
    ref = torch.nn.functional.layer_norm(x, (dl,), gam, bet, 1e-5)
    torch.cuda.synchronize()
    assert (y - ref).abs().max().item() < 1e-3
    assert torch.equal(y8, F.quantize_fp8(y, 16.0).view(torch.uint8))
    assert (y.abs() * 16.0 > 448).any()                               # (the saturating branch is exercised)


def test_encoder_at_dispatch_scale_bf16_and_fp8_track_the_fp32_mode():
    """RoBERTa-large layer geometry (hidden 1,024, 16 heads, FFN 4,096), 2 layers, 192 sequences x 64 tokens = 12,288 token rows: the size at which the
    packed projection and the FFN launches reach the eight-phase 256 x 256 kernels (gemm_p8.h EPI 2 / EPI 4: 576 / 768 tiles), with bf16-only / e4m3-only
    operand copies and the bf16 attention kernel - against the fp32 mode of the same encoder on the same weights and tokens (the mode the fixtures and the
    oracle pin at small sizes).  Bounds of the small cases, except the fp8 maximum: over 12.6 M hidden values the tail of the e4m3 rounding noise reaches further
    than over the ~10^4 of the fixtures (measured: bf16 mean 0.0045 / max 0.034 / worst row cosine 0.99997; fp8 mean 0.070 / max 0.60 / 0.9936)."""
    c = SR.cfg(1024, 2, 16, 4096, 400, 80)
    B, S = 192, 64
    g = np.random.Generator(np.random.Philox(key=77))
    lengths = [int(x) for x in g.integers(8, S + 1, size=B)]
    ids, mask = SR.make_batch(c, B, S, lengths, seed=78)
    ref = _enc(c, "fp32")(ids.cuda(), mask.cuda()).cpu()
    valid = mask.bool()
    for prec, mean_max, abs_max in (("bf16", 1e-2, 6e-2), ("fp8", 0.1, 1.0)):
        got = _enc(c, prec)(ids.cuda(), mask.cuda()).cpu()
        d = (got - ref)[valid]
        cos = torch.nn.functional.cosine_similarity(got[valid], ref[valid], dim=-1).min().item()
        print(f"{prec} encoder at 12,288 rows vs fp32 mode: mean |d| {d.abs().mean():.4f}, max |d| {d.abs().max():.4f}, worst row cosine {cos:.5f}")
        assert torch.isfinite(got[valid]).all()
        assert d.abs().mean().item() < mean_max and d.abs().max().item() < abs_max, (prec, d.abs().mean().item(), d.abs().max().item())
        assert cos > 0.99, (prec, cos)
