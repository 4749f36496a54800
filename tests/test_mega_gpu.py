"""The strip-dataflow persistent kernels (csrc/mega.hip) against the launch-list path they replace.

Both paths run the same arithmetic in the same order (same MFMA k order, same epilogue formula, same dropout RNG), so in
bf16 mode the persistent forward / backward kernels must reproduce the launch lists BIT FOR BIT: logits, loss and every
parameter gradient.  Any lost hand-off between workgroups (a stale read, a missed dependency) shows up as a difference -
the comparison is repeated so that different dispatch timings are sampled.  The launch-list path itself is pinned to the
oracle / the reference's fixtures by tests/test_model_gpu.py.
"""
import os

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_mega_library():
    """The persistent kernels are parked: the default library is built without them (csrc/Makefile, `make mega`)."""
    from mer_amd import runtime
    if "mega" not in os.path.basename(runtime.LIB_PATH) and "prof" not in os.path.basename(runtime.LIB_PATH):
        pytest.skip("persistent kernels are parked: build `make -C multimodal-emotion-recognition_amd/csrc mega` and run with "
                    "M2F_LIB=.../libm2fnet_hip_mega.so")
    if os.environ.get("M2F_ATTN_BWD_OSLAB") != "1" or os.environ.get("M2F_ATTN_BF16") != "0" or os.environ.get("M2F_SKINNY") != "0":
        pytest.skip("run with M2F_ATTN_BWD_OSLAB=1 M2F_ATTN_BF16=0 M2F_SKINNY=0: since round 3 the launch lists' attention kernels sum delta "
                    "in another order (no O slab in LDS), stage from the bf16 shadows in bf16 mode, and the classifier's [T, n_classes] "
                    "problems run as FMA kernels (skinny.hip) - the persistent kernels repeat the round-2 forms those switches bring back")


def _model(cfg, sd, mega, precision="bf16"):
    from mer_amd.model import M2FNet
    m = M2FNet(cfg, precision=precision, shape_buckets=False)      # exact shapes: partial strips / odd T are the point here
    m.load_state_dict(sd)
    m = m.to("cuda:0").train()
    m._want_mega = mega
    return m


def _step(model, batch, use_graph=False):
    text, audio, key_pad, emotion = batch
    os.environ["M2F_MEGA"] = "1" if model._want_mega else "0"          # read when a plan is built (first step)
    loss = model.train_step(text, audio, key_pad, emotion, use_graph=use_graph)
    torch.cuda.synchronize()
    eng = model.engine()
    plan = next(iter(eng.plans.values()))
    plan.check_status()
    return plan.loss.clone(), plan.logits.clone(), eng.flat_grad.clone(), plan


def _cases():
    full = synth._cfg
    return {
        "tiny_ragged": synth.CASES["tiny_ragged"][:4],
        "tiny_shared_norm": synth.CASES["tiny_shared_norm"][:4],
        "tiny_no_fam": synth.CASES["tiny_no_fam"][:4],
        "tiny_audio_only": synth.CASES["tiny_audio_only"][:4],
        "c2_slice": synth.CASES["c2_slice"][:4],
        # partial strips (T = 21), two 16-row attention tiles (L = 24), dialogues straddling 64-token strips
        "odd_T": (full(48, 64, 64, 4, 4, 4, 1, 1, 1), 3, 7, [7, 2, 5]),
        "L24": (full(96, 128, 64, 4, 8, 4, 1, 2, 2), 7, 24, [24, 3, 17, 24, 9, 1, 20]),
        "c3_slice": (full(768, 1024, 768, 8, 8, 8, 1, 1, 2), 8, 16, None),
    }


@pytest.mark.parametrize("name", sorted(_cases()))
@pytest.mark.parametrize("dropout", [0.0, 0.4])
def test_persistent_kernels_equal_launch_lists_bit_for_bit(name, dropout):
    cfg, B, L, lengths = _cases()[name]
    cfg = dict(cfg, dropout=dropout)
    sd = synth.make_state_dict(cfg)
    batch = [t.cuda() for t in synth.make_inputs(cfg, B, L, lengths, "randn")]
    torch.manual_seed(5)
    ref = _model(cfg, sd, mega=False)
    torch.manual_seed(5)                                   # same dropout RNG state in both engines
    new = _model(cfg, sd, mega=True)
    l0, z0, g0, p0 = _step(ref, batch)
    assert p0.persistent() == 0
    for rep in range(3):
        l1, z1, g1, p1 = _step(new, batch)
        assert p1.persistent() == 3, "the persistent kernels did not engage"
        if rep == 0:
            assert torch.equal(z0, z1), f"logits differ: max {(z0 - z1).abs().max().item()}"
            assert torch.equal(l0[:3], l1[:3])
            bad = (g0 != g1)
            assert not bad.any(), f"{int(bad.sum())} gradient elements differ, max {(g0 - g1).abs().max().item()}"
        if dropout == 0.0:                                 # the RNG state advances per step: only p = 0 repeats exactly
            assert torch.equal(z0, z1) and torch.equal(g0, g1)
    os.environ.pop("M2F_MEGA", None)


@pytest.mark.parametrize("workload,dropout", [("c2", 0.0), ("c2", 0.4), ("c3", 0.0), ("c3", 0.4)])
def test_persistent_kernels_full_bench_geometry_and_graph_replay(workload, dropout):
    """BASELINE C2 / C3 at full size (6+6+5 layers; B=32 / 64 x L=16): persistent kernels == launch lists, eager and as
    hipGraph, step after step (with dropout the two engines share the RNG seed and advance it in lock step)."""
    import bench
    wl = bench.WORKLOADS[workload]
    cfg, B, L = dict(wl["cfg"], dropout=dropout), wl["B"], wl["L"]
    sd = synth.make_state_dict(cfg)
    batch = list(bench.synthetic_batch(cfg, B, L, 0, "cuda:0", ragged=True))
    torch.manual_seed(11)
    ref = _model(cfg, sd, mega=False)
    torch.manual_seed(11)
    new = _model(cfg, sd, mega=True)
    for step in range(8):                                  # eager first, then capture + replays
        l0, z0, g0, _ = _step(ref, batch, use_graph=step > 0)
        l1, z1, g1, p1 = _step(new, batch, use_graph=step > 0)
        assert p1.persistent() == 3
        assert torch.equal(z0, z1), (step, (z0 - z1).abs().max().item())
        assert torch.equal(l0[:3], l1[:3]), step
        assert torch.equal(g0, g1), (step, int((g0 != g1).sum()), (g0 - g1).abs().max().item())
    os.environ.pop("M2F_MEGA", None)
