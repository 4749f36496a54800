"""Rank program of the multi-process GPU tests (tests/test_dp_multirank_gpu.py).  NOT a test module: it is started by
tests/conftest.py as  `python -m torch.distributed.run --nproc-per-node 2 tests/dp_worker.py <out_dir>`  BEFORE the pytest
process touches the GPU (a process that has initialised the GPU must not start programs on this pool), runs the scenarios below
with one process per rank and leaves one `rank<r>.pt` per rank in <out_dir> for the tests to check against single-process runs.

Backend: RCCL ("nccl") with one GPU per rank when the box has at least two GPUs; on a one-GPU box the two ranks share cuda:0
and exchange through gloo (host-staged sums, mer_amd/dp.py) - the arithmetic and the control flow of the data-parallel path
are the same, only the transport differs."""
import json
import os
import sys
import traceback

import numpy as np
import pandas as pd
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "src"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def synthetic_dataset(n_dia, d_t, d_a, seed):
    """MELD-shaped table + embedding rows (same construction as tests/test_train_loop_gpu.py::_dataset)."""
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
    lab = table["Emotion"].map(ds.EMOTIONS).to_numpy()
    text[np.arange(len(rows)), lab] += 3.0
    return ds.Dataset("train", text_embeddings=text, audio_embeddings=audio, table=table)


def loop_config(tmp_dir, dropout=0.0):
    """config of the training-loop scenario (also used by the single-process reference run in the test)."""
    import synth
    from utils import AttrDict, get_config
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        cfg = AttrDict(dict(get_config()))
    finally:
        os.chdir(cwd)
    cfg.model = AttrDict(synth._cfg(40, 48, 64, 4, 4, 4, 1, 1, 1, dropout=dropout))
    cfg.solver = AttrDict(dict(cfg.solver, epochs=6, lr=2e-3,
                               early_stopping=AttrDict(enabled=True, patience=1, restore_best_weights=True),
                               scheduler=AttrDict(enabled=False, scheduler_fn="ExponentialLR", gamma=0.9)))
    cfg.checkpoint = AttrDict(save_path=os.path.join(tmp_dir, "ck", "m2fnet.pth"), load_path=os.path.join(tmp_dir, "ck", "m2fnet.pth"),
                              save_checkpoint=True, load_checkpoint=False)
    return cfg


def run_training_loop(cfg, device, world, rank, precision="fp32", exchange="fp32", overlap=None):
    """2+ epochs over 7 dialogues in global batches of 3 (the last batch holds ONE dialogue: rank 1's shard is empty),
    validation over 5 dialogues in batches of 2 (3 batches: rank 0 scores two, rank 1 one), early stopping with patience 1."""
    import dataset as ds
    import train as tr
    from mer_amd import dp
    d_train, d_val = synthetic_dataset(7, 48, 40, 11), synthetic_dataset(5, 48, 40, 12)
    gen = torch.Generator().manual_seed(3)
    dl_train = torch.utils.data.DataLoader(d_train, collate_fn=ds.collate_fn, batch_size=3, shuffle=True, generator=gen)
    dl_val = torch.utils.data.DataLoader(d_val, collate_fn=ds.collate_fn, batch_size=2, shuffle=False)
    if world > 1:
        dl_train = dp.ShardedLoader(dl_train, rank, world)
    torch.manual_seed(0)
    model = tr.M2FNet(cfg.model, precision=precision).to(device)
    crit = tr.M2FCrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)
    opt = tr.FusedAdam(model, lr=cfg.solver.lr, weight_decay=cfg.solver.weight_decay)
    if world > 1:
        model.dp_step = dp.DataParallelStep(model, opt, n_buckets=3, exchange=exchange, overlap=overlap)
    hist = tr.training_loop(model, dl_train, dl_val, crit, opt, None, 0, cfg, device)
    torch.cuda.synchronize()
    return {"history": hist, "params": model.flat_parameters().detach().cpu().clone()}


def main():
    out_dir = sys.argv[1]
    from mer_amd import dp
    import synth
    from mer_amd.model import M2FNet
    from mer_amd.optim import FusedAdam
    world = int(os.environ["WORLD_SIZE"])
    n_gpu = torch.cuda.device_count()
    backend = "nccl" if n_gpu >= world else "gloo"
    os.environ["M2F_DIST_BACKEND"] = backend
    rank, world, local = dp.init_distributed(backend)
    device = torch.device("cuda", local if backend == "nccl" else 0)
    torch.cuda.set_device(device)
    res = {"backend": backend, "world": world, "rank": rank, "n_gpu": n_gpu}

    # ---- A: three data-parallel steps on a sharded global batch (dropout off) ------------------------------------
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    batch = synth.make_inputs(cfg, B, L, lengths, kind)
    mine = dp.shard_dialogues(B, rank, world)
    keep = max(lengths[i] for i in mine)
    shard = [t[mine][:, :keep].contiguous().to(device) for t in batch]
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="fp32").to(device).train()
    m.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
    opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
    step = dp.DataParallelStep(m, opt, n_buckets=3, exchange="fp32")
    res["A_losses"] = [float(step(*shard, use_graph=False)) for _ in range(3)]
    torch.cuda.synchronize()
    res["A_params"] = m.flat_parameters().detach().cpu().clone()

    # ---- B: same weights, same batch, dropout on: the ranks must draw DIFFERENT masks -----------------------------
    cfg_d = dict(cfg, dropout=0.4)
    torch.manual_seed(0)
    md = M2FNet(cfg_d, precision="fp32").to(device).train()
    md.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
    full = [t.to(device) for t in batch]
    md.train_step(*full, use_graph=False)
    torch.cuda.synchronize()
    res["B_rng"] = md.engine().rng.cpu().clone()
    res["B_grads"] = md.flat_gradients().detach().cpu().clone()

    # ---- D: bf16 mode, exchange overlapped with the encoders' backward (step in two parts) vs exchange after the whole backward --
    res["D"] = {}
    for overlap in (True, False):
        for exchange in ("fp32", "bf16"):
            torch.manual_seed(0)
            mb = M2FNet(cfg, precision="bf16").to(device).train()
            mb.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
            ob = FusedAdam(mb, lr=1e-3, weight_decay=0.01)
            sb = dp.DataParallelStep(mb, ob, n_buckets=3, exchange=exchange, overlap=overlap)
            losses = [float(sb(*shard, use_graph=(i > 0))) for i in range(4)]
            torch.cuda.synchronize()
            plan = next(iter(mb.engine().plans.values()))
            fresh = mb.engine().shadows_fresh()           # (before flat_parameters(): handing the buffer out invalidates)
            res["D"][(overlap, exchange)] = {"losses": losses, "params": mb.flat_parameters().detach().cpu().clone(),
                                             "split": plan.split_offset(), "exchange": sb.reducer.exchange, "fresh": fresh}
    # ... and the same steps with per-plan shadows re-cast at every forward (round-2 behaviour): the bucket-wise shadow-writing
    # optimizer must leave the same parameters
    os.environ["M2F_SHARED_SHADOWS"] = "0"
    for exchange in ("fp32", "bf16"):
        torch.manual_seed(0)
        mb = M2FNet(cfg, precision="bf16").to(device).train()
        mb.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
        ob = FusedAdam(mb, lr=1e-3, weight_decay=0.01)
        sb = dp.DataParallelStep(mb, ob, n_buckets=3, exchange=exchange, overlap=True)
        losses = [float(sb(*shard, use_graph=(i > 0))) for i in range(4)]
        torch.cuda.synchronize()
        fresh = mb.engine().shadows_fresh()
        res["D"][("recast", exchange)] = {"losses": losses, "params": mb.flat_parameters().detach().cpu().clone(), "fresh": fresh}
    os.environ.pop("M2F_SHARED_SHADOWS", None)

    # ---- E: the direct exchange (reduce-scatter + all-gather per bucket) against the all-reduce: two ranks add a + b either way --
    res["E"] = {}
    for algorithm in ("all_reduce", "rs_ag"):
        for overlap, exchange in ((False, "fp32"), (True, "bf16"), (False, "bf16")):
            torch.manual_seed(0)
            mb = M2FNet(cfg, precision="bf16").to(device).train()
            mb.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
            ob = FusedAdam(mb, lr=1e-3, weight_decay=0.01)
            sb = dp.DataParallelStep(mb, ob, n_buckets=3, exchange=exchange, overlap=overlap, algorithm=algorithm)
            losses = [float(sb(*shard, use_graph=(i > 0))) for i in range(3)]
            torch.cuda.synchronize()
            res["E"][(algorithm, overlap, exchange)] = {"losses": losses, "params": mb.flat_parameters().detach().cpu().clone(),
                                                        "g16": bool(sb.reducer.buf16_filled)}
    # ... and the bf16 exchange with the rounding pass over the fp32 buffer (M2F_GRAD_BF16=0) instead of bf16 gradients written by the step
    torch.manual_seed(0)
    mb = M2FNet(cfg, precision="bf16").to(device).train()
    mb.load_state_dict({k: v.to(device) for k, v in synth.make_state_dict(cfg).items()})
    ob = FusedAdam(mb, lr=1e-3, weight_decay=0.01)
    sb = dp.DataParallelStep(mb, ob, n_buckets=3, exchange="bf16", overlap=False, grad_bf16=False)
    losses = [float(sb(*shard, use_graph=(i > 0))) for i in range(3)]
    torch.cuda.synchronize()
    res["E"][("rounding_pass", False, "bf16")] = {"losses": losses, "params": mb.flat_parameters().detach().cpu().clone(), "g16": bool(sb.reducer.buf16_filled)}

    # ---- F: the training loop in bf16 mode with the overlapped bf16 exchange: the last global batch leaves rank 1's shard EMPTY,
    # and the empty rank must issue the same collectives (tail, fusion / classifier bucket, encoder buckets) as rank 0 ----------
    res["F"] = {}
    for overlap in (True, False):
        cfg_f = loop_config(os.path.join(out_dir, f"loop_bf16_{int(overlap)}"))
        res["F"][overlap] = run_training_loop(cfg_f, device, world, rank, precision="bf16", exchange="bf16", overlap=overlap)

    # ---- C: the training loop of src/train.py under two ranks -----------------------------------------------------
    cfg_loop = loop_config(os.path.join(out_dir, "loop_dp"))
    res["C"] = run_training_loop(cfg_loop, device, world, rank)
    res["C_files"] = sorted(os.listdir(os.path.dirname(cfg_loop.checkpoint.save_path))) if rank == 0 else []

    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception:                                       # leave the traceback where the test can show it
        with open(os.path.join(sys.argv[1], f"error_rank{os.environ.get('RANK', '0')}.txt"), "w") as f:
            f.write(traceback.format_exc())
        raise
