"""Two REAL rank processes on the device (tests/dp_worker.py, started by conftest.py before this process touches the GPU):
RCCL with one GPU per rank when the box has two, otherwise both ranks on cuda:0 exchanging through gloo.  What the
world_size-2 gloo tests on the CPU (test_dp_cpu.py) cannot cover: the HIP step under a real second rank - sharded global
batch against the single-process step on the union batch, rank-dependent dropout masks, and the whole training loop of
src/train.py (sharded loader with an EMPTY shard, validation split over the ranks, early stopping, checkpoints from rank 0)
against the single-process loop."""
import os
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "src"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import synth  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.dp2]


def test_two_rank_step_equals_single_process_step_on_the_union_batch(dp2_results):
    from mer_amd.model import M2FNet
    from mer_amd.optim import FusedAdam
    (r0, r1), _ = dp2_results
    assert r0["world"] == r1["world"] == 2 and {r0["rank"], r1["rank"]} == {0, 1}
    assert r0["backend"] == ("nccl" if r0["n_gpu"] >= 2 else "gloo")
    # both replicas took the same three steps
    assert r0["A_losses"] == r1["A_losses"] and torch.equal(r0["A_params"], r1["A_params"])
    cfg, B, L, lengths, kind = synth.CASES["tiny_ragged"]
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, kind)]
    torch.manual_seed(0)
    m = M2FNet(cfg, precision="fp32").to("cuda").train()
    sd = synth.make_state_dict(cfg)
    m.load_state_dict({k: v.cuda() for k, v in sd.items()})
    start = m.flat_parameters().detach().cpu().clone()
    opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
    losses = []
    for _ in range(3):
        losses.append(float(m.train_step(*batch, use_graph=False)))
        opt.step()
    torch.cuda.synchronize()
    ref = m.flat_parameters().detach().cpu()
    # global-denominator loss of the sharded batch == mean-over-valid loss of the union batch
    assert max(abs(a - b) for a, b in zip(r0["A_losses"], losses)) < 1e-5, (r0["A_losses"], losses)
    rel = float(((r0["A_params"] - ref).double().norm() / (ref - start).double().norm()))
    assert rel < 1e-4, rel                                  # (fp32: summation order of the two shards only)


def test_ranks_draw_different_dropout_masks(dp2_results):
    from mer_amd import dp
    (r0, r1), _ = dp2_results
    assert r0["B_rng"][:2].tolist() != r1["B_rng"][:2].tolist()
    assert r0["B_rng"][2:].tolist() == r1["B_rng"][2:].tolist()          # same step counter
    lo, hi = dp.dropout_seed(0, 1)
    assert [x & 0xFFFFFFFF for x in r1["B_rng"][:2].tolist()] == [lo, hi]
    g0, g1 = r0["B_grads"], r1["B_grads"]
    assert torch.isfinite(g0).all() and torch.isfinite(g1).all()
    # same weights, same batch, different masks: the gradients differ by much more than rounding
    assert float((g0 - g1).norm() / g0.norm()) > 0.05


def test_training_loop_under_two_ranks_equals_the_single_process_loop(dp2_results, tmp_path):
    import dp_worker as W
    (r0, r1), out_dir = dp2_results
    h0, h1 = r0["C"]["history"], r1["C"]["history"]
    assert h0 == h1 and len(h0["loss_values"]) >= 2                      # same losses, same (early) stop on both ranks
    assert torch.equal(r0["C"]["params"], r1["C"]["params"])              # replicas stayed identical
    assert len(h0["loss_values"]) < 6, "early stopping (patience 1) should have ended the run"
    assert r0["C_files"] == ["m2fnet.pth"], r0["C_files"]                # best_weights.pth restored + removed, by rank 0 only
    cfg = W.loop_config(str(tmp_path))
    ref = W.run_training_loop(cfg, torch.device("cuda:0"), 1, 0)
    assert len(ref["history"]["loss_values"]) == len(h0["loss_values"])
    for a, b in zip(ref["history"]["loss_values"] + ref["history"]["val_loss_values"], h0["loss_values"] + h0["val_loss_values"]):
        assert abs(a - b) < 5e-4 * max(1.0, abs(a)), (ref["history"], h0)
    rel = float((r0["C"]["params"] - ref["params"]).double().norm() / ref["params"].double().norm())
    assert rel < 1e-3, rel
    ck_dp = torch.load(os.path.join(out_dir, "loop_dp", "ck", "m2fnet.pth"), weights_only=False)
    ck_1 = torch.load(cfg.checkpoint.save_path, weights_only=False)
    assert ck_dp["epoch"] == ck_1["epoch"] and list(ck_dp["model_state_dict"]) == list(ck_1["model_state_dict"])


def test_overlapped_exchange_equals_exchange_after_backward(dp2_results):
    """bf16 mode under two ranks: DataParallelStep runs the step in two parts (m2f_step_part) and sends the fusion stack's and the
    classifier's gradients while the encoders' backward still runs.  Same sums, bucketed differently: losses and parameters after
    four steps (eager, capture, replays) are bit for bit those of the exchange after the whole backward, on both ranks, for the
    fp32 and the bf16 exchange."""
    (r0, r1), _ = dp2_results
    for exchange in ("fp32", "bf16"):
        a, b = r0["D"][(True, exchange)], r0["D"][(False, exchange)]
        assert a["split"] > 0 and b["split"] > 0                       # the plan can be split; `overlap=False` just does not use it
        assert a["losses"] == b["losses"], (a["losses"], b["losses"])
        assert torch.equal(a["params"], b["params"])
        assert torch.equal(a["params"], r1["D"][(True, exchange)]["params"])
        assert a["losses"][-1] < a["losses"][0]
        # the optimizer stepped bucket by bucket through its shadow-writing kernel (buckets = whole tensors): the bf16 parameter shadows
        # are current, the next forward casts nothing - and the parameters are those of the path that re-casts at every forward
        c = r0["D"][("recast", exchange)]
        assert a["fresh"] and b["fresh"] and not c["fresh"]
        assert a["losses"] == c["losses"] and torch.equal(a["params"], c["params"])


def test_reduce_scatter_all_gather_exchange_equals_the_all_reduce(dp2_results):
    """`GradReducer(algorithm="rs_ag")`: per bucket a reduce-scatter into the rank's 1/W shard and an all-gather back (SURVEY section 5's
    direct exchange) instead of one all-reduce.  Two ranks add the same two numbers either way: bit-identical steps, with and without
    the overlapped first bucket, fp32 and bf16 exchange."""
    (r0, r1), _ = dp2_results
    for overlap, exchange in ((False, "fp32"), (True, "bf16"), (False, "bf16")):
        a, b = r0["E"][("all_reduce", overlap, exchange)], r0["E"][("rs_ag", overlap, exchange)]
        assert a["losses"] == b["losses"], (a["losses"], b["losses"])
        assert torch.equal(a["params"], b["params"])
        assert torch.equal(b["params"], r1["E"][("rs_ag", overlap, exchange)]["params"])
        assert a["losses"][-1] < a["losses"][0]
    # bf16 exchange, whole-step form: the step left its gradients in the exchange buffer itself (m2f_plan_grad_bf16) - same bits as the
    # rounding pass over the fp32 buffer it replaces
    a, c = r0["E"][("all_reduce", False, "bf16")], r0["E"][("rounding_pass", False, "bf16")]
    assert a["g16"] and not c["g16"]
    assert a["losses"] == c["losses"] and torch.equal(a["params"], c["params"])


def test_empty_shard_follows_the_overlapped_bucket_schedule(dp2_results):
    """bf16 mode, bf16 exchange, `overlap=True`, and a last global batch with ONE dialogue: rank 1 holds nothing and has never
    built a plan for that shape.  It must still issue the collectives rank 0 issues - tail, the fusion stack's / classifier's
    bucket, then the encoder buckets cut from [0, split) - or the sums pair up wrongly (or RCCL hangs).  The run ends, both
    replicas hold the same parameters, and they are those of the run that exchanges after the whole backward."""
    (r0, r1), _ = dp2_results
    on0, on1, off0 = r0["F"][True], r1["F"][True], r0["F"][False]
    assert on0["history"] == on1["history"] and torch.equal(on0["params"], on1["params"])
    assert len(on0["history"]["loss_values"]) >= 2
    assert on0["history"] == off0["history"]
    assert torch.equal(on0["params"], off0["params"])
