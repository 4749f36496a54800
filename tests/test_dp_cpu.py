"""world_size-2 gloo tests (CPU) of the data-parallel exchange: the [gradients | den | num] sum-all-reduce with the
GLOBAL valid-utterance denominator reproduces the single-process mean-over-valid loss and gradients exactly,
whereas averaging per-rank means does not.  Local gradients come from the CPU oracle (the HIP path needs a GPU)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import synth
    import mer_amd  # noqa: F401
    from mer_amd import dp, layout
    from oracle import m2fnet_oracle as O
    r, w, _ = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    sd = synth.make_state_dict(cfg)
    text, audio, key_pad, emotion = synth.make_inputs(cfg, B, L, lengths, kind)
    mine = dp.shard_dialogues(B, rank, world)
    t, a, kp, em = text[mine], audio[mine], key_pad[mine], emotion[mine]
    # local SUM-gradient: grad of (mean loss * local den) = what m2f_step(normalise=0) produces
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = O.forward(leaves, cfg, t, a, kp)
    den = (em != -1).sum().to(torch.float32)
    local_mean = O.cross_entropy(logits, em)
    (local_mean * den).backward()
    specs, total = layout.param_specs(layout.M2FConfig.from_model_config(cfg))
    buf = torch.zeros(total + dp.TAIL)
    for sp in specs:
        buf[sp.offset: sp.offset + sp.numel] = leaves[sp.name].grad.reshape(-1)
    red = dp.GradReducer(buf, total, n_buckets=3)
    red.set_loss_terms(den, local_mean.detach() * den)
    red.all_reduce()
    g = buf[:total] / red.global_den
    out = {"loss": float(red.global_loss()), "den": float(red.global_den), "local_mean": float(local_mean)}
    if rank == 0:
        _, ref_loss, ref_grads = O.loss_and_grads(sd, cfg, text, audio, key_pad, emotion)
        worst = 0.0
        for sp in specs:
            ref = ref_grads[sp.name].reshape(-1)
            worst = max(worst, float((g[sp.offset: sp.offset + sp.numel] - ref).abs().max() / ref.abs().max().clamp_min(1e-6)))
        out.update(ref_loss=float(ref_loss), worst=worst, n_valid=int((emotion != -1).sum()))
    # pipelined exchange + update: buckets reduced tail-first, each "optimizer" range waits only for its own bucket
    buf2 = torch.full((1000 + dp.TAIL,), float(rank + 1))
    red2 = dp.GradReducer(buf2, 1000, n_buckets=4)
    seen = []

    class FakeOpt:
        def step_ranges(self, ranges, before_each=None, grads=None):
            for i, (lo, hi) in enumerate(ranges):
                before_each(i)
                seen.append((lo, hi, float(buf2[lo]), float(buf2[min(hi, buf2.numel()) - 1])))
    red2.reduce_and_step(FakeOpt())
    total = float(sum(range(1, world + 1)))
    # bf16 exchange: parameter gradients travel as bf16 (own buffer handed to the optimizer), the tail stays fp32
    buf3 = torch.cat([torch.full((1000,), 0.1 * (rank + 1)), torch.full((dp.TAIL,), float(rank + 1))])
    red3 = dp.GradReducer(buf3, 1000, n_buckets=3, exchange="bf16")
    seen3 = []

    class FakeOpt16:
        def step_ranges(self, ranges, before_each=None, grads=None):
            for i, (lo, hi) in enumerate(ranges):
                before_each(i)
                seen3.append((lo, hi, grads.dtype, float(grads[lo]), float(grads[hi - 1]), float(buf3[1000 + 1])))
    red3.reduce_and_step(FakeOpt16())
    want16 = float(sum(torch.tensor(0.1 * (r + 1)).to(torch.bfloat16).float() for r in range(world)))
    out["bf16_ok"] = (len(seen3) == 3 and all(d == torch.bfloat16 and abs(a - want16) < 2e-2 and abs(b - want16) < 2e-2 and t == total
                                             for _, _, d, a, b, t in seen3)
                      and sorted((lo, hi) for lo, hi, *_ in seen3) == sorted(red3.param_chunks)
                      and float(buf3[0]) == float(torch.tensor(0.1 * (rank + 1))))       # fp32 gradients left untouched
    out["pipelined_ok"] = (all(a == total and b == total for _, _, a, b in seen) and seen[0][1] == buf2.numel()
                           and sorted((lo, hi) for lo, hi, _, _ in seen) == sorted(red2.chunks))
    # the direct exchange (reduce-scatter + all-gather per bucket) gives the sums of the all-reduce, fp32 and bf16
    rs_ok = True
    for exchange in ("fp32", "bf16"):
        sums = {}
        for algorithm in dp.ALGORITHMS:
            g5 = torch.Generator().manual_seed(5 + rank)
            b5 = torch.cat([torch.randn(4096, generator=g5), torch.full((dp.TAIL,), float(rank + 1))])
            red5 = dp.GradReducer(b5, 4096, n_buckets=3, exchange=exchange, algorithm=algorithm)
            got = []

            class Opt5:
                def step_ranges(self, ranges, before_each=None, grads=None):
                    for i, (lo, hi) in enumerate(ranges):
                        before_each(i)
                        got.append((lo, hi, (grads if grads is not None else b5)[lo:min(hi, 4096)].float().clone()))
            red5.reduce_and_step(Opt5())
            sums[algorithm] = (got, float(b5[4096 + 1]))
        a, b = sums["all_reduce"], sums["rs_ag"]
        rs_ok = rs_ok and a[1] == b[1] == total and len(a[0]) == len(b[0]) == 3
        rs_ok = rs_ok and all(x[:2] == y[:2] and torch.equal(x[2], y[2]) for x, y in zip(a[0], b[0]))
    out["rs_ag_ok"] = rs_ok
    # split exchange with an EMPTY rank: rank 1 contributes zeros and has nothing to run between the first bucket and the rest, but
    # issues the same collectives - the sums pair up bucket by bucket
    for exchange in ("fp32", "bf16"):
        b6 = torch.cat([(torch.arange(2048) % 128).float() / 16.0,           # (exact in bf16)
                        torch.tensor([0.0, 3.0, 6.0]), torch.zeros(dp.TAIL - 3)])
        red6 = dp.GradReducer(b6, 2048, n_buckets=3, exchange=exchange)
        if rank == 1:
            red6.zero_contribution()
        got6, ran = [], []

        class Opt6:
            def step_ranges(self, ranges, before_each=None, grads=None):
                for i, (lo, hi) in enumerate(ranges):
                    before_each(i)
                    got6.append((lo, hi, (grads if grads is not None else b6)[lo:min(hi, 2048)].float().clone()))
        red6.reduce_and_step_split(Opt6(), (lambda: ran.append(1)) if rank == 0 else (lambda: None), 1280)
        want = (torch.arange(2048) % 128).float() / 16.0
        ok6 = got6[0][:2] == (1280, 2048 if exchange == "bf16" else b6.numel()) and sorted(x[0] for x in got6)[0] == 0
        ok6 = ok6 and all(torch.equal(x[2], want[x[0]:min(x[1], 2048)]) for x in got6) and float(b6[2048 + 1]) == 3.0
        out["empty_split_ok_" + exchange] = bool(ok6 and (ran == [1] if rank == 0 else ran == []))
    means = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(means, local_mean.detach().reshape(1))
    out["mean_of_means"] = float(torch.stack(means).mean())
    t = dp.reduce_metrics([1.0 + rank])
    out["max_time"] = t[0]
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_global_denominator():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    r0 = res[0]
    assert r0["den"] == r0["n_valid"]
    assert abs(r0["loss"] - r0["ref_loss"]) < 1e-5 and abs(res[1]["loss"] - r0["ref_loss"]) < 1e-5
    assert r0["worst"] < 1e-4, r0["worst"]
    assert abs(r0["mean_of_means"] - r0["ref_loss"]) > 1e-4, "ranks hold different valid counts: mean of means differs"
    assert r0["max_time"] == 2.0 and res[1]["max_time"] == 2.0
    assert r0["pipelined_ok"] and res[1]["pipelined_ok"]
    assert r0["bf16_ok"] and res[1]["bf16_ok"]
    assert r0["rs_ag_ok"] and res[1]["rs_ag_ok"]
    for r in (0, 1):
        assert res[r]["empty_split_ok_fp32"] and res[r]["empty_split_ok_bf16"], res[r]


def test_shard_dialogues_partition():
    import mer_amd  # noqa: F401
    from mer_amd import dp
    for n, w in ((32, 8), (33, 4), (5, 2), (3, 8)):
        parts = [dp.shard_dialogues(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


def test_reducer_buckets_cover_buffer():
    import mer_amd  # noqa: F401
    from mer_amd import dp
    buf = torch.zeros(10_000 + dp.TAIL)
    for nb in (1, 3, 8, 64):
        red = dp.GradReducer(buf, 10_000, n_buckets=nb)
        assert red.chunks[0][0] == 0 and red.chunks[-1][1] == buf.numel()
        assert all(a[1] == b[0] for a, b in zip(red.chunks[:-1], red.chunks[1:]))


def test_bench_starts_its_own_ranks_and_refuses_a_wrong_world_size():
    """`python bench.py --gpus 2` without a launcher environment must start two ranks itself (torch.distributed.run children)
    and form a 2-rank group; a launcher environment whose world size disagrees with --gpus must exit non-zero."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(29900 + os.getpid() % 90)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"rendezvous": 2, "backend": "gloo"}
    bad = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=bad,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr


def test_sharded_loader_partitions_every_batch_and_yields_empty_shards():
    import mer_amd  # noqa: F401
    from mer_amd import dp
    g = torch.Generator().manual_seed(3)
    batches = []
    for B, L in ((5, 7), (2, 4), (1, 3)):
        emo = torch.randint(0, 7, (B, L), generator=g)
        for b in range(B):
            emo[b, int(torch.randint(1, L + 1, (1,), generator=g)):] = -1
        emo[0, :] = emo[0, :].clamp_min(0)              # the batch's longest dialogue is full length
        batches.append({"text": torch.randn(B, L, 6, generator=g), "audio": torch.randn(B, L, 4, generator=g),
                        "emotion": emo, "padding_mask": emo == -1})
    world = 3
    shards = [list(dp.ShardedLoader(batches, r, world)) for r in range(world)]
    assert all(len(s) == len(batches) for s in shards)
    for i, full in enumerate(batches):
        B = full["emotion"].shape[0]
        seen = 0
        for r in range(world):
            sh = shards[r][i]
            mine = dp.shard_dialogues(B, r, world)
            assert sh["emotion"].shape[0] == len(mine)
            if not mine:
                assert sh["text"].shape[0] == 0 and sh["padding_mask"].shape[0] == 0        # empty shard, still yielded
                continue
            Ls = sh["emotion"].shape[1]
            assert torch.equal(sh["emotion"], full["emotion"][mine][:, :Ls])
            assert torch.equal(sh["text"], full["text"][mine][:, :Ls]) and torch.equal(sh["audio"], full["audio"][mine][:, :Ls])
            assert (full["emotion"][mine][:, Ls:] == -1).all(), "only all-padding columns may be dropped"
            assert (~sh["padding_mask"]).any(dim=0).all() or Ls == 1
            seen += int((sh["emotion"] != -1).sum())
        assert seen == int((full["emotion"] != -1).sum())
    # a rank with an empty shard contributes exact zeros to the exchange
    buf = torch.randn(100 + dp.TAIL)
    red = dp.GradReducer(buf, 100)
    red.zero_contribution()
    assert float(buf.abs().sum()) == 0.0 and float(red.global_den) == 0.0
