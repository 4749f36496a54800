"""GPU test of the drop-in driver surface (src/train.py, src/test.py, src/dataset.py): training_loop / train / validate /
test with the reference's signatures, on a synthetic MELD-shaped dataset (the real CSVs are not in the container)."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "src"))

import synth  # noqa: E402
from oracle import m2fnet_oracle as O  # noqa: E402


def _dataset(n_dia, d_t, d_a, seed):
    import dataset as ds
    g = np.random.default_rng(seed)
    rows = []
    for d in range(n_dia):
        for u in range(int(g.integers(1, 10))):
            rows.append((f"utt {d}-{u}", list(ds.EMOTIONS)[int(g.integers(0, 7))], d, u))
    order = g.permutation(len(rows))
    table = pd.DataFrame([rows[i] for i in order], columns=["Utterance", "Emotion", "Dialogue_ID", "Utterance_ID"])
    text = torch.from_numpy(g.standard_normal((len(rows), d_t)).astype(np.float32))
    audio = torch.from_numpy(g.standard_normal((len(rows), d_a)).astype(np.float32))
    # make the label learnable from the text embedding so the loss must fall
    lab = table["Emotion"].map(ds.EMOTIONS).to_numpy()
    text[np.arange(len(rows)), lab] += 3.0
    return ds.Dataset("train", text_embeddings=text, audio_embeddings=audio, table=table)


def test_training_loop_checkpoint_validate_and_test(tmp_path, monkeypatch):
    monkeypatch.chdir(ROOT)
    import dataset as ds
    import train as tr
    import test as te
    from utils import AttrDict, get_config
    cfg = AttrDict(dict(get_config()))
    model_cfg = synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1, dropout=0.1)
    cfg.model = AttrDict(model_cfg)
    cfg.solver = AttrDict(dict(cfg.solver, epochs=4, lr=2e-3,
                               early_stopping=AttrDict(enabled=True, patience=2, restore_best_weights=True),
                               scheduler=AttrDict(enabled=True, scheduler_fn="ExponentialLR", gamma=0.9)))
    cfg.checkpoint = AttrDict(save_path=str(tmp_path / "ck" / "m2fnet.pth"), load_path=str(tmp_path / "ck" / "m2fnet.pth"),
                              save_checkpoint=True, load_checkpoint=False)
    d_train, d_val = _dataset(40, 48, 40, 1), _dataset(12, 48, 40, 2)
    dl_train = torch.utils.data.DataLoader(d_train, collate_fn=ds.collate_fn, batch_size=8, shuffle=True)
    dl_val = torch.utils.data.DataLoader(d_val, collate_fn=ds.collate_fn, batch_size=8, shuffle=False)
    device = torch.device("cuda:0")
    torch.manual_seed(0)
    model = tr.M2FNet(cfg.model).to(device)
    crit = tr.M2FCrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)
    opt = tr.FusedAdam(model, lr=cfg.solver.lr, weight_decay=cfg.solver.weight_decay)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
    out = tr.training_loop(model, dl_train, dl_val, crit, opt, sched, 0, cfg, device)
    losses = out["loss_values"]
    assert len(losses) >= 2 and losses[-1] < losses[0] - 0.2, losses
    ck = torch.load(cfg.checkpoint.save_path)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict"}
    assert list(ck["model_state_dict"].keys()) == list(synth.make_state_dict(model_cfg).keys())
    assert abs(opt.param_groups[0]["lr"] - cfg.solver.lr * 0.9 ** len(losses)) < 1e-9

    # validate(): per-batch metrics, unweighted mean over batches == the oracle's rule on the same logits
    loss_v, acc, f1 = tr.validate(model, dl_val, crit, device)
    model.eval()
    batches = []
    with torch.inference_mode():
        for b in dl_val:
            lg = model(b["text"].to(device), b["audio"].to(device), b["padding_mask"].to(device)).cpu()
            batches.append((lg, b["emotion"]))
    acc_o, f1_o = O.epoch_metrics(batches)
    assert abs(acc - acc_o) < 1e-9 and abs(f1 - f1_o) < 1e-9 and np.isfinite(loss_v)
    assert 0.0 <= acc <= 1.0 and 0.0 <= f1 <= 1.0
    acc_t, f1_t = te.test(model, dl_val, device)
    assert abs(acc_t - acc) < 1e-9 and abs(f1_t - f1) < 1e-9

    # non-fused loop body (model(), criterion, backward, step) trains too
    model2 = tr.M2FNet(cfg.model).to(device)
    opt2 = tr.FusedAdam(model2, lr=2e-3, weight_decay=0.01)
    monkeypatch.setattr(tr, "_step_mode", lambda m, c: (False, True))
    l0 = tr.train(model2, dl_train, crit, opt2, 0, False, device)
    l1 = tr.train(model2, dl_train, crit, opt2, 1, False, device)
    assert l1 < l0


def test_device_loader_yields_the_dataloader_batches():
    """runtime.device_batcher: the HBM-resident loader produces exactly the batches of DataLoader + collate_fn (same
    dialogue order without shuffling, same tensors, last batch partial) and a permutation of them when shuffling."""
    import dataset as ds
    d = _dataset(21, 16, 8, 7)
    ref = list(torch.utils.data.DataLoader(d, collate_fn=ds.collate_fn, batch_size=8, shuffle=False))
    dev = ds.DeviceLoader(d, batch_size=8, shuffle=False, device="cuda", num_workers=2, pin_memory=True)
    got = list(dev)
    assert len(dev) == len(ref) == len(got) == 3
    for a, b in zip(got, ref):
        for k in ("text", "audio", "padding_mask", "emotion"):
            assert a[k].is_cuda and torch.equal(a[k].cpu(), b[k]), k
    sh = ds.DeviceLoader(d, batch_size=8, shuffle=True, device="cuda", seed=3)
    e1 = torch.cat([b["emotion"][~b["padding_mask"]] for b in sh]).cpu()
    e2 = torch.cat([b["emotion"][~b["padding_mask"]] for b in sh]).cpu()
    all_lab = torch.cat([b["emotion"][~b["padding_mask"]] for b in ref])
    assert sorted(e1.tolist()) == sorted(all_lab.tolist()) == sorted(e2.tolist())      # every utterance once per epoch
    assert e1.tolist() != e2.tolist()                                                  # new permutation each epoch


def test_collate_fn_contract_matches_oracle():
    import dataset as ds
    d = _dataset(6, 16, 8, 5)
    items = [d[i] for i in range(len(d))]
    a, b = ds.collate_fn(items), O.collate(items)
    for k in ("text", "audio", "padding_mask", "emotion"):
        assert torch.equal(a[k], b[k]), k
    assert a["emotion"].dtype == torch.int64 and a["padding_mask"].dtype == torch.bool


def test_plan_cache_is_bounded_over_real_dialogue_lengths():
    """MELD batches have their longest dialogue anywhere in 1..33 plus a partial last batch: the engine rounds shapes up
    to buckets (L to multiples of 16, B to 1/2/4/8/16/24/...), so L in 3..33 needs three train plans, the plan cache is
    an LRU of at most `max_plans`, and the bucketed run computes exactly what the exact-shape run computes."""
    from mer_amd.model import M2FNet
    cfg = synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1)
    sd = synth.make_state_dict(cfg)
    m = M2FNet(cfg)                                         # shape buckets on (default)
    m.load_state_dict(sd)
    m = m.to("cuda:0").train()
    exact = M2FNet(cfg, shape_buckets=False)
    exact.load_state_dict(sd)
    exact = exact.to("cuda:0").train()
    g = torch.Generator().manual_seed(0)
    for L in list(range(3, 34, 3)) + [33, 16, 17]:
        B = int(torch.randint(3, 9, (1,), generator=g))
        lengths = [int(x) for x in torch.randint(1, L + 1, (B,), generator=g)]
        lengths[0] = L
        batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn", seed=L)]
        loss = m.train_step(*batch, use_graph=True)
        if L in (3, 17, 33):                                # bucketed == exact, bit for bit (the filler slots add exact zeros)
            ref = exact.train_step(*batch, use_graph=False)
            eng, eng_x = m.engine(), exact.engine()
            plan = eng.plans[next(reversed(eng.plans))]
            plan_x = eng_x.plans[next(reversed(eng_x.plans))]
            valid = ~batch[2]
            assert plan.logits.shape == plan_x.logits.shape == (B, L, 7)
            assert torch.equal(plan.logits[valid], plan_x.logits[valid])
            assert abs(loss.item() - ref.item()) < 1e-6
            assert (eng.flat_grad - eng_x.flat_grad).abs().max().item() <= 1e-6 * eng_x.flat_grad.abs().max().item()
    eng = m.engine()
    shapes = sorted({(k[0], k[1]) for k in eng.plans})
    # (forced packing - M2F_PACKED=1 - adds the token-row bucket to the plan key: more plans of the same (B, L) buckets)
    assert len(eng.plans) <= (6 if os.environ.get("M2F_PACKED") != "1" else 16) and {s[1] for s in shapes} <= {16, 32, 48}, shapes
    assert len({s[1] for s in shapes}) <= 3
    assert eng.plan_bytes() <= eng.max_plans * max(p.nbytes() for p in eng.plans.values())
    # LRU: with room for two plans only, the oldest goes
    eng.max_plans = 2
    for L in (5, 20, 40, 5):
        batch = [t.cuda() for t in synth.make_inputs(cfg, 4, L, None, "randn", seed=L)]
        m.train_step(*batch, use_graph=True)
        assert len(eng.plans) <= 2
    torch.cuda.synchronize()


def _grads_of(m):
    return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()


def test_two_forwards_in_one_bucket_then_backward_of_the_first():
    """Reference order the plan cache must allow: `out_a = model(A); out_b = model(B); loss_a.backward()` with A and
    B of different exact shapes in ONE shape bucket (L = 5 and L = 9 -> 16).  The second forward gets a plan instance of its
    own instead of overwriting the activations the first graph still needs; both backward passes give the gradients of a
    run that saw only that batch."""
    from mer_amd.model import M2FNet
    cfg = synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1, dropout=0.0)
    sd = synth.make_state_dict(cfg)

    def fresh():
        m = M2FNet(cfg)
        m.load_state_dict(sd)
        return m.to("cuda:0").train()

    a = [t.cuda() for t in synth.make_inputs(cfg, 4, 5, [5, 2, 3, 1], "randn", seed=1)]
    b = [t.cuda() for t in synth.make_inputs(cfg, 4, 9, [9, 4, 1, 7], "randn", seed=2)]
    crit = torch.nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)

    def loss_of(m, batch):
        return crit(m(batch[0], batch[1], batch[2]).permute(0, 2, 1), batch[3])

    ref = {}
    for name, batch in (("a", a), ("b", b)):
        m = fresh()
        loss_of(m, batch).backward()
        ref[name] = _grads_of(m)
    m = fresh()
    la = loss_of(m, a)
    lb = loss_of(m, b)
    eng = m.engine()
    assert len(eng.plans) == 2 and len({k[:-1] for k in eng.plans}) == 1          # one bucket, two instances
    la.backward()
    assert torch.equal(_grads_of(m), ref["a"])
    lb.backward()
    assert torch.equal(_grads_of(m), ref["b"])
    # both graphs are done: the next forward of the bucket re-uses instance 0, no third plan appears
    del la, lb
    loss_of(m, a).backward()
    assert len(eng.plans) == 2 and torch.equal(_grads_of(m), ref["a"])


def test_busy_plan_survives_eviction_and_closed_plan_raises():
    """`_evict` must not close a plan whose activations a live autograd graph still needs (it used to: the later backward
    dereferenced a destroyed plan); a closed plan raises instead of handing NULL to the library."""
    from mer_amd import runtime
    from mer_amd.model import M2FNet
    cfg = synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1, dropout=0.0)
    m = M2FNet(cfg)
    m.load_state_dict(synth.make_state_dict(cfg))
    m = m.to("cuda:0").train()
    eng = m.engine()
    eng.max_plans = 1
    a = [t.cuda() for t in synth.make_inputs(cfg, 4, 5, None, "randn", seed=1)]
    c = [t.cuda() for t in synth.make_inputs(cfg, 4, 20, None, "randn", seed=3)]
    crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
    la = crit(m(a[0], a[1], a[2]).permute(0, 2, 1), a[3])
    pa = next(iter(eng.plans.values()))
    with torch.no_grad():
        m(c[0], c[1], c[2])                                  # another bucket: over the cap, but plan A is busy
    assert pa.handle and pa.busy()
    la.backward()                                            # ... so this still works
    assert not pa.busy()
    g = _grads_of(m)
    assert torch.isfinite(g).all() and g.abs().max() > 0
    with torch.no_grad():
        m(c[0], c[1], c[2])                                  # now A is idle and goes
    assert not pa.handle
    with pytest.raises(runtime.HipError, match="closed"):
        pa.forward()
    torch.cuda.synchronize()


def test_training_loop_with_in_loop_text_encoder(tmp_path, monkeypatch):
    """BASELINE config C5 through the drop-in (runtime.text_encoder): batches carry token ids, `move_batch` turns them into the text rows
    with the RoBERTa-shaped encoder's [CLS] states (mer_amd.roberta on the HIP kernels, inference mode), `training_loop` trains the
    fusion model on them - forward, criterion, backward, optimizer, validation, checkpoint, all unchanged."""
    monkeypatch.chdir(ROOT)
    import dataset as ds
    import train as tr
    from metrics import move_batch
    from utils import AttrDict, get_config
    cfg = AttrDict(dict(get_config()))
    model_cfg = synth._cfg(40, 64, 64, 4, 4, 4, 1, 1, 1, dropout=0.0)
    cfg.model = AttrDict(model_cfg)
    cfg.solver = AttrDict(dict(cfg.solver, epochs=3, lr=2e-3, early_stopping=AttrDict(enabled=False, patience=2, restore_best_weights=True),
                               scheduler=AttrDict(enabled=False, scheduler_fn="ExponentialLR", gamma=0.9)))
    cfg.checkpoint = AttrDict(save_path=str(tmp_path / "ck" / "m2fnet.pth"), load_path=str(tmp_path / "ck" / "m2fnet.pth"),
                              save_checkpoint=True, load_checkpoint=False)
    te_cfg = {"enabled": True, "precision": "bf16",
              "geometry": dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=60, max_position_embeddings=40)}
    S = 12

    def tokens(d, seed):                                           # the label is written into the tokens: the encoder's rows carry it
        g = torch.Generator().manual_seed(seed)
        n = len(d._labels)
        ids = torch.randint(10, 60, (n, S), generator=g)
        ids[:, 0] = 0
        ids[:, 1] = 3 + d._labels
        ids[:, 2] = 3 + d._labels
        lens = torch.randint(4, S + 1, (n,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
        ids[mask == 0] = 1
        return ids, mask
    d_train, d_val = _dataset(40, 64, 40, 1), _dataset(12, 64, 40, 2)
    for d, seed in ((d_train, 5), (d_val, 6)):
        d.token_ids, d.token_mask = tokens(d, seed)
        d.text_embeddings = torch.zeros_like(d.text_embeddings)     # the pre-extracted rows must not be what the model learns from
    dl_train = torch.utils.data.DataLoader(d_train, collate_fn=ds.collate_fn, batch_size=8, shuffle=True)
    dl_val = torch.utils.data.DataLoader(d_val, collate_fn=ds.collate_fn, batch_size=8, shuffle=False)
    device = torch.device("cuda:0")
    torch.manual_seed(0)
    model = tr.M2FNet(cfg.model).to(device)
    enc = tr.build_text_encoder(te_cfg, 64, device)
    object.__setattr__(model, "text_encoder", enc)
    assert "text_encoder" not in dict(model.named_modules()) and not any(k.startswith("text_encoder.") for k in model.state_dict())
    # what move_batch hands to the model: [CLS] rows at valid slots, zeros at pads
    batch = next(iter(dl_val))
    assert batch["text_ids"].shape[:2] == batch["padding_mask"].shape and batch["text_ids"].shape[2] == S
    text, audio, emotion, mask = move_batch(batch, device, text_encoder=enc)
    valid = ~batch["padding_mask"]
    ref = enc.cls_embeddings(batch["text_ids"][valid].to(device), batch["text_mask"][valid].to(device)).float()
    assert torch.equal(text[valid.to(device)], ref) and float(text[~valid.to(device)].abs().max() if (~valid).any() else 0.0) == 0.0
    assert text.shape[2] == 64 and torch.isfinite(text).all()
    crit = tr.M2FCrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)
    opt = tr.FusedAdam(model, lr=cfg.solver.lr, weight_decay=cfg.solver.weight_decay)
    out = tr.training_loop(model, dl_train, dl_val, crit, opt, None, 0, cfg, device)
    losses = out["loss_values"]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses
    assert np.isfinite(out["val_loss_values"]).all()
    # the same rows, pre-extracted the reference's way (text/embeddings.py:64-90 = text_context.cls_dump), train the same model the same way
    # on its first batch: the in-loop path is the two-stage path without the files
    from mer_amd.text_context import cls_dump
    items = [{"idx": torch.arange(len(d_val._labels)), "text": d_val.token_ids.to(device), "attention_mask": d_val.token_mask.to(device)}]
    dumped = cls_dump(enc, items, len(d_val._labels), str(tmp_path / "emb"), "val")
    rows = torch.as_tensor(d_val.rows[0])
    got = move_batch(ds.collate_fn([d_val[0]]), device, text_encoder=enc)[0][0, : len(rows)].cpu()
    assert (got - dumped[rows]).abs().max().item() < 2e-2          # (bf16 mode: batch composition changes nothing but the tile edges)
