"""CPU tests: the oracle (oracle/m2fnet_oracle.py) against the golden fixtures recorded from the real
reference (tests/golden/make_golden.py), and against torch / sklearn pieces the reference calls
directly (nn.CrossEntropyLoss, torch.optim.Adam, sklearn metrics)."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import m2fnet_oracle as O

CASES = list(synth.CASES)
TOL_LOGITS = 2e-5      # fp32 CPU restatement vs fp32 CPU reference (different op order only)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _inputs(name, fx):
    cfg, B, L, lengths, kind = synth.CASES[name]
    if kind == "real":
        _, _, key_pad, emotion = synth.make_inputs(cfg, B, L, lengths, "randn")
        text, audio = torch.from_numpy(fx["text"]), torch.from_numpy(fx["audio"])
    else:
        text, audio, key_pad, emotion = synth.make_inputs(cfg, B, L, lengths, kind)
    return cfg, text, audio, key_pad, emotion


@pytest.mark.parametrize("name", CASES)
def test_oracle_logits_match_reference(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, text, audio, key_pad, emotion = _inputs(name, fx)
    sd = synth.make_state_dict(cfg)
    logits = O.forward(sd, cfg, text, audio, key_pad)
    valid = ~key_pad
    err = (logits - torch.from_numpy(fx["logits_eval"])).abs()[valid].max().item()
    assert err < TOL_LOGITS, err
    err_t = (logits - torch.from_numpy(fx["logits_train"])).abs()[valid].max().item()
    assert err_t < TOL_LOGITS, err_t


@pytest.mark.parametrize("name", CASES)
def test_oracle_loss_and_grads_match_reference(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, text, audio, key_pad, emotion = _inputs(name, fx)
    sd = synth.make_state_dict(cfg)
    _, loss, grads = O.loss_and_grads(sd, cfg, text, audio, key_pad, emotion)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5
    lw = O.cross_entropy(O.forward(sd, cfg, text, audio, key_pad), emotion, synth.CLASS_WEIGHTS)
    assert abs(lw.item() - float(fx["loss_weighted"])) < 1e-5
    names = [str(n) for n in fx["grad_names"]]
    keys = list(sd.keys())
    for j, k in enumerate(names):
        g = grads[k].double()
        ref_norm = float(fx["grad_norms"][j])
        assert abs(float(g.norm()) - ref_norm) <= 5e-4 * max(ref_norm, 1e-3), (k, float(g.norm()), ref_norm)
        probe = synth.digest_vector(tuple(g.shape), 3, keys.index(k)).double()
        assert abs(float((g * probe).sum()) - float(fx["grad_dots"][j])) <= 1e-3 * max(ref_norm, 1e-3) * 8 + 3e-5, k  # +abs: key-bias grads are pure roundoff (softmax shift invariance)
        if "grad::" + k in fx:
            ref = torch.from_numpy(fx["grad::" + k]).double()
            assert (g - ref).abs().max().item() <= 3e-5 + 5e-4 * ref.abs().max().item(), k  # abs floor: ReLU-gate flips on real rows (|text| up to 21)


@pytest.mark.parametrize("name", ["tiny_ragged", "tiny_shared_norm", "tiny_odd_heads"])
def test_oracle_fam_intermediates(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, text, audio, key_pad, _ = _inputs(name, fx)
    sd = synth.make_state_dict(cfg)
    inter = {}
    O.forward(sd, cfg, text, audio, key_pad, inter=inter)
    f0 = inter["fam0"]
    valid = ~key_pad
    assert (f0["x"] - torch.from_numpy(fx["fam0_mha_out"])).abs()[valid].max() < 2e-5
    assert (f0["y"] - torch.from_numpy(fx["fam0_out"])).abs()[valid].max() < 2e-5
    # the reference returns head-averaged attention weights (need_weights default, model.py:14)
    assert (f0["p"].mean(dim=1) - torch.from_numpy(fx["fam0_attn_avg"])).abs().max() < 1e-6


@pytest.mark.parametrize("name", ["tiny_ragged", "tiny_shared_norm", "c1"])
def test_oracle_adam_three_steps(golden_dir, name):
    fx = _load(golden_dir, name)
    cfg, text, audio, key_pad, emotion = _inputs(name, fx)
    sd = synth.make_state_dict(cfg)
    uniq, order = {}, []
    for k, v in sd.items():
        if id(v) not in uniq:
            uniq[id(v)] = k
            order.append(k)
    params = [sd[k] for k in order]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    losses = []
    for step in range(1, 4):
        _, loss, grads = O.loss_and_grads(sd, cfg, text, audio, key_pad, emotion)
        losses.append(loss.item())
        O.adam_step(params, [grads[k] for k in order], m, v, step, lr=1e-3, weight_decay=0.01)
    assert np.allclose(losses, fx["adam_losses"], rtol=0, atol=3e-5), (losses, fx["adam_losses"])
    norms = np.array([float(p.double().norm()) for p in params])
    assert np.allclose(norms, fx["adam3_norms"], rtol=2e-5, atol=1e-6)
    logits = O.forward(sd, cfg, text, audio, key_pad)
    assert (logits - torch.from_numpy(fx["adam3_logits_eval"])).abs()[~key_pad].max() < 2e-4


def test_oracle_shipped_depth_on_all_real_validation_rows(golden_dir):
    """c2p_real_val: the reference's config.yaml model verbatim (6 + 6 encoder layers, 5 FAM layers, 768^3) on ALL 1,108 matched
    rows of the real val.pkl embeddings, in the validation loader's batches: the oracle reproduces the reference's logits,
    every prediction, and the validation rule of src/train.py:245-272 (per-batch loss / accuracy / weighted-F1, plain mean)."""
    from sklearn.metrics import accuracy_score, f1_score
    fx = _load(golden_dir, "c2p_real_val")
    cfg, sd, batches = synth.c2p_real_val_case(fx)
    ref = torch.from_numpy(fx["logits"])
    losses, accs, f1s, worst = [], [], [], 0.0
    for bi, (text, audio, key_pad, emotion, rows) in enumerate(batches):
        lg = O.forward(sd, cfg, text, audio, key_pad)
        valid = ~key_pad
        worst = max(worst, (lg[valid] - ref[rows[valid]]).abs().max().item())
        assert torch.equal(lg[valid].argmax(1), ref[rows[valid]].argmax(1))
        losses.append(float(O.cross_entropy(lg, emotion, None, 0.1)))
        p, t = lg.argmax(2)[valid].numpy(), emotion[valid].numpy()
        accs.append(accuracy_score(t, p))
        f1s.append(f1_score(t, p, average="weighted"))
    assert worst < 1e-4, worst
    assert np.allclose(losses, fx["loss_per_batch"], atol=2e-5)
    assert np.allclose(accs, fx["acc_per_batch"], atol=1e-12) and np.allclose(f1s, fx["f1_per_batch"], atol=1e-12)
    assert abs(np.mean(f1s) - float(fx["f1"])) < 1e-12 and abs(np.mean(accs) - float(fx["acc"])) < 1e-12


def test_oracle_cross_entropy_vs_torch():
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(6, 9, 7, generator=g) * 2
    tgt = torch.randint(0, 7, (6, 9), generator=g)
    tgt[0, 4:] = -1
    tgt[3, 1:] = -1
    for w in (None, synth.CLASS_WEIGHTS):
        ref = torch.nn.CrossEntropyLoss(weight=w, ignore_index=-1, label_smoothing=0.1)(logits.permute(0, 2, 1), tgt)
        assert abs(O.cross_entropy(logits, tgt, w).item() - ref.item()) < 1e-6


def test_oracle_adam_vs_torch():
    g = torch.Generator().manual_seed(4)
    p0 = [torch.randn(5, 3, generator=g), torch.randn(7, generator=g)]
    ref_p = [torch.nn.Parameter(p.clone()) for p in p0]
    opt = torch.optim.Adam(ref_p, lr=5e-3, weight_decay=0.01)
    mine = [p.clone() for p in p0]
    m = [torch.zeros_like(p) for p in mine]
    v = [torch.zeros_like(p) for p in mine]
    for step in range(1, 6):
        grads = [torch.randn(p.shape, generator=g) for p in mine]
        for rp, gr in zip(ref_p, grads):
            rp.grad = gr.clone()
        opt.step()
        O.adam_step(mine, grads, m, v, step, lr=5e-3, weight_decay=0.01)
    for a, b in zip(mine, ref_p):
        assert (a - b.detach()).abs().max() < 1e-6


def test_oracle_metrics_fixture(golden_dir):
    fx = _load(golden_dir, "metrics")
    batches = [(torch.from_numpy(fx[f"logits{b}"]), torch.from_numpy(fx[f"emotion{b}"])) for b in range(4)]
    for b, (lg, em) in enumerate(batches):
        acc, f1 = O.batch_metrics(lg, em)
        assert abs(acc - fx["acc_per_batch"][b]) < 1e-12 and abs(f1 - fx["f1_per_batch"][b]) < 1e-12
    acc, f1 = O.epoch_metrics(batches)
    assert abs(acc - float(fx["acc"])) < 1e-12 and abs(f1 - float(fx["f1"])) < 1e-12


def test_oracle_collate_contract():
    d = [{"text": torch.ones(3, 4), "audio": torch.ones(3, 2), "emotion": [torch.tensor([1]), torch.tensor([0]), torch.tensor([6])]},
         {"text": 2 * torch.ones(1, 4), "audio": 2 * torch.ones(1, 2), "emotion": [torch.tensor([5])]}]
    b = O.collate(d)
    assert b["text"].shape == (2, 3, 4) and b["audio"].shape == (2, 3, 2)
    assert b["emotion"].tolist() == [[1, 0, 6], [5, -1, -1]]
    assert b["padding_mask"].tolist() == [[False, False, False], [False, True, True]]
    assert b["text"][1, 1:].abs().sum() == 0 and b["emotion"].dtype == torch.int64


def test_pad_contents_do_not_change_valid_logits(golden_dir):
    """valid-position logits are independent of what padded slots hold."""
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    sd = synth.make_state_dict(cfg)
    text, audio, key_pad, _ = synth.make_inputs(cfg, B, L, lengths, kind)
    a = O.forward(sd, cfg, text, audio, key_pad)
    t2, a2 = text.clone(), audio.clone()
    t2[key_pad] = 123.0
    a2[key_pad] = -77.0
    b = O.forward(sd, cfg, t2, a2, key_pad)
    assert (a - b)[~key_pad].abs().max() == 0


@pytest.mark.skipif(not os.path.exists("/root/reference/src/model.py"), reason="reference only in the build container")
def test_oracle_vs_live_reference_default_init():
    """Live check on the reference's OWN default init (not the synthetic weights)."""
    import sys
    import types
    import warnings
    sys.path.insert(0, "/root/reference/src")
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore")
    import model as ref_model
    cfg, B, L, lengths, kind = synth.CASES["tiny_odd_heads"]

    def ns(d):
        return types.SimpleNamespace(**{k: (ns(v) if isinstance(v, dict) else v) for k, v in d.items()})
    torch.manual_seed(5)
    m = ref_model.M2FNet(ns(cfg)).eval()
    text, audio, key_pad, _ = synth.make_inputs(cfg, B, L, lengths, kind)
    with torch.inference_mode():
        ref = m(text, audio, key_pad)
    mine = O.forward(dict(m.state_dict()), cfg, text, audio, key_pad)
    assert (mine - ref)[~key_pad].abs().max() < TOL_LOGITS
