#!/usr/bin/env python
"""Benchmark of the M2FNet training step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one optimizer step on one batch of synthetic dialogues already resident in HBM:
forward + criterion + backward (one hipGraph launch), gradient all-reduce over RCCL when N > 1, fused Adam.
The headline metric (BASELINE.json) is utterances/s; per-GPU work is fixed (weak scaling, global batch = N * B).
Workload = BASELINE.json configs[2] ("c3": shipped depth, roberta-large 1024 + wav2vec2 768, B=64 x L=16, bf16) - the largest
single-GPU configuration and the one north_star's target is stated on; configs[1] ("c2") runs with --workload c2.
Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel = the grouped MFMA GEMM,
timed live with hipEvents per launch; `roofline.fam_gemm` = the same figure over the fusion-attention stack's GEMM
launches alone, the kernels north_star's target is stated on), `secondary` (the other single-GPU configuration, C2) and
`cpu_baseline` (the CPU oracle on the host cores, N=1 only).

`python bench.py --gpus N` started WITHOUT a torchrun environment launches its own N ranks (one child process per GPU
through torch.distributed.run, decided before anything touches the GPU) and exits with their status; a rank whose RCCL
world size differs from --gpus exits non-zero instead of measuring something else.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mer_amd  # noqa: E402,F401
from mer_amd import dp, layout, runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402
from mer_amd.optim import FusedAdam  # noqa: E402


def model_cfg(d_a, d_t, d_f, h_a, h_t, h_f, nl, nf, dropout=0.4):
    return {"dropout": dropout,
            "AUDIO": {"enabled": True, "embedding_size": d_a, "n_head": h_a, "n_transformers": 1, "n_encoder_layers": nl},
            "TEXT": {"enabled": True, "embedding_size": d_t, "n_head": h_t, "n_transformers": 1, "n_encoder_layers": nl},
            "FAM": {"enabled": True, "embedding_size": d_f, "n_head": h_f, "n_layers": nf},
            "CLASSIFIER": {"hidden_size": 768, "output_size": 7, "n_layers": 2}}


# BASELINE.json configs.  audio_mel is 300-d: n_head must divide 300 (5 -> head_dim 60).
WORKLOADS = {
    "c1": dict(cfg=model_cfg(512, 768, 768, 8, 8, 8, 1, 1), B=4, L=16, name="C1 1+1 layers 768/512/768 B4xL16"),
    "c2": dict(cfg=model_cfg(300, 768, 768, 5, 8, 8, 6, 5), B=32, L=16,
               name="C2 M2FNet full (6 enc layers/modality, 5 FAM) roberta-base 768 + audio_mel 300, B32xL16"),
    "c2p": dict(cfg=model_cfg(768, 768, 768, 8, 8, 8, 6, 5), B=32, L=16, name="C2' shipped config.yaml 768/768/768 B32xL16"),
    "c3": dict(cfg=model_cfg(768, 1024, 768, 8, 8, 8, 6, 5), B=64, L=16, name="C3 roberta-large 1024 + wav2vec2 768, B64xL16"),
    "c3b256": dict(cfg=model_cfg(768, 1024, 768, 8, 8, 8, 6, 5), B=256, L=16, name="C3 geometry at 4x the batch (B256xL16; not a BASELINE config: shows what the fixed cost per launch hides at B64)"),
    "c3l24": dict(cfg=model_cfg(768, 1024, 768, 8, 8, 8, 6, 5), B=64, L=24, name="C3 roberta-large 1024 + wav2vec2 768, B64xL24 (SURVEY 8-d secondary length)"),
}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md


def synthetic_batch(cfg, B, L, rank, device, ragged=False):
    """text = 0.63*randn, audio = 0.23*randn, labels randint(0,7); headline = all dialogues full length,
    secondary (`ragged`) = MELD-like lengths clamp(round(N(9.6, 5)), 1, L), pads zeroed, labels -1 on pads."""
    g = torch.Generator().manual_seed(1234 + rank)
    text = torch.randn(B, L, cfg["TEXT"]["embedding_size"], generator=g) * 0.63
    audio = torch.randn(B, L, cfg["AUDIO"]["embedding_size"], generator=g) * 0.23
    emotion = torch.randint(0, 7, (B, L), generator=g)
    mask = torch.zeros(B, L, dtype=torch.bool)
    if ragged:
        lengths = torch.clamp(torch.round(torch.randn(B, generator=g) * 5.0 + 9.6), 1, L).to(torch.int64)
        mask = torch.arange(L)[None, :] >= lengths[:, None]
        text[mask] = 0.0
        audio[mask] = 0.0
        emotion[mask] = -1
    return text.to(device), audio.to(device), mask.to(device), emotion.to(device)


def cpu_baseline(cfg, B, L, budget_s=20.0):
    """The CPU oracle (explicit-op restatement, verified == reference) on this host's cores: fwd + criterion +
    backward + Adam on the same synthetic workload; bounded sample."""
    from oracle import m2fnet_oracle as O
    torch.manual_seed(0)
    c = layout.M2FConfig.from_model_config(cfg)
    specs, _ = layout.param_specs(c)
    sd = {}
    for sp in specs:
        if sp.alias_of:
            sd[sp.name] = sd[sp.alias_of]
        elif sp.kind in ("ln_w",):
            sd[sp.name] = torch.ones(sp.shape)
        elif sp.kind in ("ln_b", "attn_in_b", "attn_out_b"):
            sd[sp.name] = torch.zeros(sp.shape)
        else:
            sd[sp.name] = (torch.rand(sp.shape) * 2 - 1) / max(sp.fan_in, 1) ** 0.5
    text, audio, mask, emotion = synthetic_batch(cfg, B, L, 0, "cpu")
    uniq = {}
    for k, v in sd.items():
        uniq.setdefault(id(v), k)
    order = list(uniq.values())
    params = [sd[k] for k in order]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    threads = torch.get_num_threads()
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        _, _, grads = O.loss_and_grads(sd, cfg, text, audio, mask, emotion)
        step += 1
        O.adam_step(params, [grads[k] for k in order], m, v, step, lr=5e-5, weight_decay=0.01)
        times.append(time.perf_counter() - t0)
        if (time.perf_counter() - t_start > budget_s and len(times) >= 3) or len(times) >= 40:
            break
    steady = times[1:] if len(times) > 1 else times
    sec = sum(steady) / len(steady)
    return {"value": B * L / sec, "unit": "utterances/s", "cores": threads, "kind": "port",
            "sample": f"{len(steady)} timed steps (+1 warm-up) of the same workload (fwd+CE+bwd+Adam, fp32, dropout off), "
                      f"{sec * 1e3:.0f} ms/step, torch threads={threads}, os.cpu_count()={os.cpu_count()}"}



def source_hash():
    """sha256 over the kernel sources and headers: committed profile digests carry the hash they were measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multimodal-emotion-recognition_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load_profile_digest(kind, workload, dtype):
    """profiles/<kind>_<workload>_<dtype>.json if present; `stale` when it was measured on other kernel sources."""
    path = os.path.join(ROOT, "profiles", f"{kind}_{workload}_{dtype}.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    d["_source"] = os.path.relpath(path, ROOT)
    d["_stale"] = d.get("source_hash") != source_hash()
    return d


LAUNCH_NAMES = ["gemm_fwd", "gemm_dgrad", "gemm_wgrad", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "dropout", "ce", "ln_reduce",
                "cast_bf16", "persistent_fwd", "persistent_bwd"]
PARTS = ["encoders", "fusion", "classifier"]


def run_workload(wl, dtype, rank, world, device, steps, warmup, use_graph, ragged, buckets, exchange, roofline=True, dump="", packed=False,
                 repeats=1, overlap=False, algorithm="all_reduce", fused_adam=False, grad_bf16=True, fp32_grad_leg=True):
    cfg, B, L = wl["cfg"], wl["B"], wl["L"]
    torch.manual_seed(0)                               # identical replicas on every rank
    model = M2FNet(cfg, precision=dtype, shape_buckets=False).to(device).train()      # the plan IS the workload's (B, L): no bucket padding
    opt = FusedAdam(model, lr=5e-5, weight_decay=0.01)
    stepper = dp.DataParallelStep(model, opt, n_buckets=buckets, exchange=exchange, overlap=overlap, algorithm=algorithm)
    text, audio, mask, emotion = synthetic_batch(cfg, B, L, rank, device, ragged=ragged)
    n_valid = int((~mask).sum().item())                # utterances of this rank's batch (= B*L unless --ragged)
    eng = model.engine()
    # --packed: token rows = valid utterances only (m2f_plan_create_packed); pays off with --ragged
    plan = eng.plan(B, L, True, True, n_valid if packed else None)

    split = plan.split_offset() if (world > 1 and overlap) else 0
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):
        # inputs are resident in the plan's staging buffers before the timed region starts
        plan.set_inputs(text, audio, mask, emotion)

        fused_steps = [0]

        def one_step():
            # bf16 mode, N = 1: the fused Adam kernel also writes the bf16 parameter shadows, the forward then skips its parameter
            # casts (the engine compares the parameters' version counters: any other write to them brings the casts back)
            plan.params_fresh(eng.shadows_fresh())
            if fused_adam and world == 1 and opt.prepare_fused(plan):
                # round 4: the optimizer step INSIDE the step's graph - the weight-gradient launch applies Adam to the elements whose
                # gradient it holds in registers (dW never reaches memory), one more launch updates biases / LayerNorm parameters
                plan.step(0.1, False, False, use_graph)
                opt.finish_fused(plan)
                fused_steps[0] += 1
                return
            if split:
                # N > 1: the step in two parts - the fusion stack's / classifier's gradient bucket (the tail of the flat buffer, final
                # after part 0) is on the wire while the encoders' backward (part 1) runs; the encoder buckets follow, each bucket's
                # fused-Adam launch behind its own collective
                plan.step_part(0, 0.1, False, False, use_graph)
                stepper.reducer.reduce_and_step_split(opt, lambda: plan.step_part(1, 0.1, False, False, use_graph), split)
                return
            plan.step(0.1, False, False, use_graph)                     # fwd + CE + bwd (sum-gradient; tail <- den, num)
            stepper.reducer.reduce_and_step(opt)      # RCCL all-reduce buckets (tail first) pipelined with fused Adam

        if world > 1:
            # bf16 exchange: the plan was created before the stepper pointed the engine at the exchange buffer - arm it now
            eng._arm_grad_bf16(plan)
            stepper.reducer.buf16_filled = (not split) and stepper.reducer.buf16 is not None and getattr(plan, "_g16_ref", None) is stepper.reducer.buf16
        g16_on = False
        if world == 1 and grad_bf16 and dtype == "bf16":
            # N = 1 with the gradient precision of the bf16 exchange at N > 1: rounded once to bf16 by the step (the weight-gradient launch
            # writes bf16 dW), read as bf16 by the optimizer - no fp32 dW round trip
            g16_on = model.set_grad_bf16(True)
        eng.publish_grads()
        for _ in range(max(warmup, 3)):                # >= 3: eager warm-up, graph capture, first replay
            one_step()
        side.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        side.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        shadows_fresh_after_loop = bool(eng.shadows_fresh())          # (N > 1: the bucket-wise optimizer launches write the shadows too)
        elapsed = dp.reduce_metrics([elapsed], device=device)[0]
        loss = float(stepper.reducer.global_loss().item())
        plan.check_status()
        # the same bracket four more times (spread of the measurement; `value` stays the FIRST bracket = exactly `steps` steps)
        rep_ms = [elapsed / steps * 1e3]
        for _ in range(repeats - 1):
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            side.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            rep_ms.append(dp.reduce_metrics([time.perf_counter() - tr0], device=device)[0] / steps * 1e3)

        # ---- N > 1: what the exchange costs, so that a scaling record explains itself ---------------------------------------------
        comm = None
        if world > 1:
            red = stepper.reducer
            n_c = max(5, min(steps, 20))

            def bracket(fn, n):
                torch.distributed.barrier()
                torch.cuda.synchronize()
                tc0 = time.perf_counter()
                for _ in range(n):
                    fn()
                side.synchronize()
                torch.cuda.synchronize()
                torch.distributed.barrier()
                return dp.reduce_metrics([time.perf_counter() - tc0], device=device)[0] / n * 1e3
            for _ in range(3):
                red.exchange_only()
            alone_ms = bracket(red.exchange_only, n_c)          # the step's collectives alone: same buckets, dtypes and order, no compute
            red.stub = True                                    # the same step on identical data with every collective a no-op
            for _ in range(3):
                one_step()
            nocomm_ms = bracket(one_step, n_c)
            red.stub = False
            one_step()                                         # (leave the replicas in step with each other again: real sums)
            comm = {"algorithm": red.algorithm, "exchange_dtype": red.exchange, "buckets": len(red.param_chunks if red.exchange == "bf16" else red.chunks),
                    "overlap_with_backward": bool(split), "gradients_rounded_by": "the step (bf16 dW from the weight-gradient launch)" if red.buf16_filled else ("a pass over the fp32 buffer" if red.exchange == "bf16" else None),
                    "bytes_per_step": red.bytes_per_step(),
                    "allreduce_alone_ms": alone_ms, "algbw_GBps_alone": red.bytes_per_step() / (alone_ms * 1e-3) / 1e9,
                    "step_without_collectives_ms": nocomm_ms, "exposed_ms": elapsed / steps * 1e3 - nocomm_ms, "steps_each": n_c,
                    "note": "exposed_ms = ms_per_step - the same step with the collectives stubbed to no-ops on identical data; "
                            "allreduce_alone_ms = the same buckets with no compute around them (max over ranks)"}

        # ---- secondary figure: forward + criterion + backward only (SURVEY 8-d's strict metric; `value` above also
        # pays for the optimizer and, at N > 1, the gradient exchange) ---------------------------------------------
        n_fb = max(10, min(steps, 50))
        fp32_grad_ms = None
        if g16_on and fp32_grad_leg:
            # ... and the same step with fp32 gradients (rounds 1-3), on the same clock, beside it
            model.set_grad_bf16(False)
            for _ in range(3):
                one_step()
            side.synchronize()
            tg0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            side.synchronize()
            fp32_grad_ms = (time.perf_counter() - tg0) / steps * 1e3
            model.set_grad_bf16(True)                      # (what follows measures the default mode again)
            for _ in range(2):
                one_step()
        for _ in range(3):
            plan.step(0.1, False, False, use_graph)
        side.synchronize()
        tf0 = time.perf_counter()
        for _ in range(n_fb):
            plan.step(0.1, False, False, use_graph)
        side.synchronize()
        fb_sec = (time.perf_counter() - tf0) / n_fb

        # ---- roofline of the dominant kernel (grouped GEMM), per-launch hipEvent timing ------------------
        for _ in range(5):
            plan.step_timed(0.1, False, False)
        reps = [plan.step_timed(0.1, False, False) for _ in range(10)]
        ev_empty_ms, ev_trivial_ms = runtime.event_overhead(200)
    n_l = len(reps[0])
    avg_ms = [sum(r[i][1] for r in reps) / len(reps) for i in range(n_l)]
    kinds = [reps[0][i][0] & 31 for i in range(n_l)]
    parts = [reps[0][i][0] >> 5 for i in range(n_l)]
    flops = [reps[0][i][2] for i in range(n_l)]
    if dump:
        with open(dump, "w") as f:
            for i in range(n_l):
                tf = flops[i] / (avg_ms[i] * 1e-3) / 1e12 if avg_ms[i] > 0 else 0.0
                f.write(f"{i:4d} {LAUNCH_NAMES[kinds[i]]:14s} {PARTS[parts[i]]:10s} {avg_ms[i] * 1e3:9.2f} us {flops[i] / 1e9:9.3f} GFLOP {tf:8.1f} TFLOP/s\n")
    # every launch that carries GEMM FLOPs: the grouped GEMM launches
    gemm_idx = [i for i in range(n_l) if kinds[i] in (0, 1, 2)]
    gemm_ms = sum(avg_ms[i] for i in gemm_idx)
    gemm_fl = sum(flops[i] for i in gemm_idx)
    fam_idx = [i for i in gemm_idx if kinds[i] in (0, 1) and parts[i] == 1]
    fam_ms, fam_fl = sum(avg_ms[i] for i in fam_idx), sum(flops[i] for i in fam_idx)
    c = model.m2f_config
    _, fb_per_slot = layout.flops_per_slot(c, L)
    ms_per_step = elapsed / steps * 1e3
    nv = torch.tensor([float(n_valid)], dtype=torch.float64, device=device)
    if world > 1:                                      # SUM over ranks of the valid-utterance counts
        torch.distributed.all_reduce(nv)
    n_valid_all = int(nv.item())
    utt_per_s = n_valid_all / (elapsed / steps)
    slots_per_s = world * (plan.T if plan.packed else B * L) / (elapsed / steps)          # padded slots are computed too (unless packed)
    achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = PEAK_TFLOPS[dtype]
    wl_key = [k for k, v in WORKLOADS.items() if v is wl][0]
    # HBM-side bytes per GEMM launch and the kernels' own begin..end durations cannot be read from inside the process: they
    # come from committed rocprofv3 passes of this same command (tools/traffic.sh, tools/kstats_bench.sh) - and are only
    # quoted while the kernel sources are the ones they were measured on
    traffic = load_profile_digest("traffic", wl_key, dtype)
    kstats = load_profile_digest("kstats", wl_key, dtype)
    rocprof = None
    if kstats is not None:
        rocprof = {"source": kstats["_source"], "stale": kstats["_stale"]}
        if not kstats["_stale"]:
            rocprof.update({"avg_launch_us": kstats["avg_launch_us"], "gemm_ms_per_step": kstats["gemm_ms_per_step"],
                            "achieved": kstats.get("gemm_gflop_per_step", gemm_fl / 1e9) / kstats["gemm_ms_per_step"], "unit": "TFLOP/s"})   # GFLOP / ms
    out = {
        "metric": "utterances/sec (fwd+bwd) M2FNet fusion, MELD dialogues, 1/2/4/8 MI355X",
        "value": utt_per_s, "unit": "utterances/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": wl["name"] + (" [MELD-like ragged lengths, valid utterances counted]" if ragged else ""),
                   "dialogues_per_gpu": B, "max_utt": L, "global_batch_dialogues": world * B, "valid_utterances": n_valid_all,
                   "d_text": c.d_text, "d_audio": c.d_audio, "d_fam": c.d_fam, "params": layout.param_count(c),
                   "step": (("fwd+CE+bwd+Adam (1 hipGraph: the weight-gradient launch applies the optimizer in its epilogue)" if fused_steps[0] else "fwd+CE+bwd (1 hipGraph)") if not split else "fwd+CE+bwd in two hipGraphs") +
                           (f" + {'RCCL' if torch.distributed.get_backend() == 'nccl' else torch.distributed.get_backend()} grad all-reduce ({stepper.reducer.exchange}" + (", fusion / classifier bucket sent under the encoders' backward)" if split else ")") if world > 1 else "") + ("" if fused_steps[0] else " + fused Adam"),
                   "gradients": ("bf16 (rounded once by the step; fp32 moments, fp32 master parameters)" if g16_on or (world > 1 and stepper.reducer.exchange == "bf16") else "fp32"),
                   "dropout": c.dropout, "parallelism": f"dp{world}", "hipgraph": use_graph,
                   "launches_per_step": plan.num_launches(),
                   "token_rows": plan.T, "plan_shape": [plan.B, plan.L], "packed": bool(plan.packed),
                   "param_shadows": "written by the fused Adam kernel (no parameter casts in the forward)" if eng.wshadow is not None and shadows_fresh_after_loop
                                    else ("re-cast at the head of every forward" if dtype == "bf16" else None),
                   "source_hash": source_hash()},
        "loss": loss,
        "repeats": {"n": len(rep_ms), "steps_each": steps, "ms_per_step": rep_ms, "median": sorted(rep_ms)[len(rep_ms) // 2],
                    "min": min(rep_ms), "max": max(rep_ms),
                    "note": "`value` / `ms_per_step` are the first bracket; the others repeat it back to back"},
        "fwd_bwd_only": {"ms_per_step": fb_sec * 1e3, "utterances_per_s_rank0": n_valid / fb_sec, "steps": n_fb,
                         "note": "fwd + CE + bwd graph replays on rank 0, optimizer and gradient exchange excluded"},
        "fp32_gradients": None if fp32_grad_ms is None else {"ms_per_step": fp32_grad_ms, "value": n_valid / (fp32_grad_ms * 1e-3), "steps": steps,
                                                               "note": "the same step with fp32 gradients between backward and optimizer (M2FNet.set_grad_bf16(False)); "
                                                                       "`value` runs with gradients rounded once to bf16, as under the bf16 exchange at N > 1"},
        "step_tflops": slots_per_s * fb_per_slot / 1e12,
        "step_frac_of_peak": slots_per_s * fb_per_slot / 1e12 / (peak * world),
    }
    if comm is not None:
        out["comm"] = comm
    if roofline:
        out["roofline"] = {
            "bound": "mfma", "kernel": "bf16: m2f_gemm16_ring_kernel (forward / input-gradient launches as 256x128, 128x128, 128x64 or 64x64 ring tiles) + m2f_gemm_p8_kernel (the weight-gradient table launch, 256x256 tiles on the eight-phase schedule); fp32: m2f_gemm_kernel - the grouped MFMA GEMM launches of one step",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": None if traffic is None or traffic["_stale"] else float(traffic["traffic_bytes_per_launch"]),
            "traffic_source": None if traffic is None else {"file": traffic["_source"], "stale": traffic["_stale"]},
            "launches_per_step": len(gemm_idx), "avg_launch_us": gemm_ms / max(len(gemm_idx), 1) * 1e3,
            # the fusion-attention stack's own GEMM launches (q/v, k, out-projection, Linear(2E->E); forward + input gradient):
            # north_star states its roofline target on these.  Their weight-gradient problems run inside the ONE table launch
            # of the step and cannot be timed apart from the other layers'.
            "fam_gemm": None if not fam_idx else {
                "achieved": fam_fl / (fam_ms * 1e-3) / 1e12, "peak": peak, "frac": fam_fl / (fam_ms * 1e-3) / 1e12 / peak,
                "launches": len(fam_idx), "gflop": fam_fl / 1e9, "ms": fam_ms, "avg_launch_us": fam_ms / len(fam_idx) * 1e3,
                "what": "forward + input-gradient GEMM launches of the fusion stack (hipEvent intervals)"},
            # what a hipEvent interval holds besides the kernel's own begin..end (which is what rocprofv3 reports):
            # NOT subtracted from `achieved`, stated so the two can be reconciled
            "event_interval_overhead_us": {"empty_pair": ev_empty_ms * 1e3, "pair_around_one_thread_kernel": ev_trivial_ms * 1e3},
            "rocprof": rocprof,
            "algorithmic_gflop_per_step": gemm_fl / 1e9, "gemm_ms_per_step": gemm_ms,
            "all_kernels_ms_per_step_eager": sum(avg_ms),
        }
    del stepper, opt, model, plan, eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("M2F_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=os.environ.get("M2F_PRECISION", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ragged", action="store_true", help="secondary workload of SURVEY 8-d: MELD-like dialogue lengths, "
                    "`value` then counts VALID utterances only")
    ap.add_argument("--packed", action="store_true", help="packed token layout (valid utterances only; with --ragged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--buckets", type=int, default=4,
                    help="gradient all-reduce buckets at N > 1 (31 MB each for the bf16 exchange at C2: large enough for "
                         "ring bandwidth, small enough that only the last bucket's Adam launch is exposed)")
    ap.add_argument("--grad-exchange", default="auto", choices=["auto", "fp32", "bf16"],
                    help="dtype of the gradient all-reduce at N > 1 (auto: bf16 for --dtype bf16, fp32 for --dtype fp32)")
    ap.add_argument("--dump-launches", default="", help="write the per-launch timing table (kind, us, GFLOP) to this file")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group (gloo when there is no GPU), check its size against --gpus, print it, exit: "
                         "the launch path without the measurement (tests/test_dp_cpu.py)")
    ap.add_argument("--repeats", type=int, default=5, help="timed brackets of `--steps` steps each (the first one is `value`)")
    ap.add_argument("--dp-overlap", action="store_true", default=os.environ.get("M2F_DP_OVERLAP", "0") == "1",
                    help="N > 1: the fusion stack's / classifier's gradient bucket travels under the encoders' backward (the step runs as two "
                         "hipGraphs).  OFF by default: verified bit for bit against the plain order through gloo staging, never yet run on RCCL "
                         "with two devices - switch it on once a multi-GPU run has compared the two")
    ap.add_argument("--no-overlap", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--dp-algorithm", default=os.environ.get("M2F_DP_ALGORITHM", "all_reduce"), choices=list(dp.ALGORITHMS),
                    help="N > 1: per bucket one all-reduce, or reduce-scatter + all-gather (SURVEY section 5's direct exchange over the "
                         "fully connected xGMI links)")
    ap.add_argument("--fused-adam", action="store_true", default=os.environ.get("M2F_FUSED_ADAM", "0") == "1",
                    help="N = 1, bf16: apply the optimizer in the weight-gradient launch's epilogue (dW never reaches memory; bit-identical "
                         "parameters).  OFF by default: measured SLOWER at C3 (fused launch 897 us against 243 + 556 us for table launch + "
                         "optimizer kernel, profiles/r04_dev_fused_adam_ab.txt) - tile-shaped 64-byte-per-row accesses to p / m / v reach "
                         "4.2 TB/s where the linear optimizer kernel streams 6.0")
    ap.add_argument("--grad-fp32", action="store_true", help="N = 1, bf16: keep fp32 gradients between the step and the optimizer (rounds 1-3).  Default "
                    "since round 4: the step leaves its gradients rounded once to bf16 - the precision every rank's gradient has under the bf16 "
                    "exchange at N > 1 - and the optimizer reads those (no fp32 dW round trip); the fp32-gradient step is timed beside it "
                    "(`fp32_gradients`)")
    ap.add_argument("--no-fp32-grad-leg", action="store_true", help="skip the fp32-gradient step timed beside the default line (profiling runs: every "
                    "kernel of the run is then the default mode's)")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the fp32 (1e-3 parity mode) leg of the same workload")
    ap.add_argument("--secondary", default="c2", choices=sorted(WORKLOADS) + ["none"],
                    help="second single-GPU configuration reported under `secondary` (N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher environment: start the ranks ourselves - fresh child processes, before this process touches the GPU
        import subprocess
        port = int(os.environ.get("MASTER_PORT", "29533"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    rank, world, local = dp.init_distributed()
    if torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()       # the communicator's size, not the environment's claim
    if world != args.gpus:
        print(f"bench.py: the process group has {world} rank(s) but --gpus {args.gpus} was asked for; refusing to report a "
              f"{args.gpus}-GPU number", file=sys.stderr)
        sys.exit(3)
    if args.rendezvous_only:
        if rank == 0:
            print(json.dumps({"rendezvous": world, "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None}))
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    n_dev = torch.cuda.device_count()
    if local >= n_dev and os.environ.get("M2F_DIST_BACKEND") == "gloo" and n_dev > 0:
        local = local % n_dev          # rehearsal only: several gloo ranks sharing one GPU (tests the N > 1 code path, not its speed)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    runtime.require_gpu()

    wl = WORKLOADS[args.workload]
    use_graph = not args.no_graph
    exchange = args.grad_exchange if args.grad_exchange != "auto" else ("bf16" if args.dtype == "bf16" else "fp32")
    res = run_workload(wl, args.dtype, rank, world, device, args.steps, args.warmup, use_graph, args.ragged, args.buckets, exchange,
                       roofline=True, dump=args.dump_launches if rank == 0 else "", packed=args.packed, repeats=max(1, args.repeats),
                       overlap=args.dp_overlap and not args.no_overlap, algorithm=args.dp_algorithm, fused_adam=args.fused_adam, grad_bf16=not args.grad_fp32, fp32_grad_leg=not args.no_fp32_grad_leg)
    if rank == 0:
        out = res
        if world == 1 and args.dtype == "bf16" and not args.no_parity_leg and not args.ragged and not args.packed:
            # the mode that meets north_star's 1e-3 logits bound (exact-fp32 MFMA, tests/test_bench_geometry_gpu.py), same workload,
            # same protocol, shorter run: what parity-exact costs, on the same clock as the headline
            par = run_workload(wl, "fp32", rank, world, device, max(10, args.steps // 5), max(3, args.warmup // 2), use_graph, False,
                               args.buckets, "fp32", roofline=True, dump="", repeats=3)
            out["parity_mode"] = {"dtype": "fp32", "value": par["value"], "unit": par["unit"], "ms_per_step": par["ms_per_step"],
                                  "steps": par["steps"], "fwd_bwd_only": par["fwd_bwd_only"], "repeats": par["repeats"],
                                  "roofline": {k: par["roofline"][k] for k in ("achieved", "peak", "frac", "unit", "launches_per_step", "fam_gemm")},
                                  "tolerance": "logits within 1e-3 of the reference (asserted < 2e-4 at this geometry)"}
        if world == 1 and args.secondary != "none" and args.secondary != args.workload:
            # the other single-GPU configuration of BASELINE.json, same protocol, shorter run
            sec = run_workload(WORKLOADS[args.secondary], args.dtype, rank, world, device, max(20, args.steps // 2), args.warmup,
                               use_graph, False, args.buckets, exchange, roofline=True, dump="")
            out["secondary"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "fwd_bwd_only", "step_tflops", "step_frac_of_peak")}
            out["secondary"]["workload"] = sec["config"]["workload"]
            out["secondary"]["roofline"] = {k: sec["roofline"][k] for k in ("achieved", "peak", "frac", "fam_gemm", "launches_per_step")}
        if world == 1 and not args.no_cpu_baseline:
            try:
                share = len(os.sched_getaffinity(0))
            except AttributeError:
                share = os.cpu_count() or 1
            torch.set_num_threads(max(1, min(share, 16)))       # the GPU box gives one GPU a 16-core share
            eval_cfg = dict(wl["cfg"], dropout=0.0)
            out["cpu_baseline"] = cpu_baseline(eval_cfg, wl["B"], wl["L"], args.cpu_budget)
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
