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
    # measured (round 4): see the message; bounds stated here
    assert stats["bf16"][0] < FP8_LOGIT_BOUNDS["bf16"] and stats["bf16"][1] >= 0.97, (stats, spread)
    assert stats["fp8"][0] < FP8_LOGIT_BOUNDS["fp8"] and stats["fp8"][1] >= 0.90, (stats, spread)
    assert stats["fp8"][0] >= stats["bf16"][0] * 0.5, stats               # fp8 is the coarser mode: the numbers must say so


FP8_LOGIT_BOUNDS = {"bf16": 5e-2, "fp8": 0.3}
