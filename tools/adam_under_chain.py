#!/usr/bin/env python
"""How much of the optimizer can hide under the encoders' backward chain?  (round 4, one GPU, C3 geometry, fp32 gradients)

  S   whole step (one graph) + optimizer over everything, one stream                      - the shipped order
  P   part 0, part 1 (two graphs) + optimizer over everything, one stream                 - what the split alone costs
  O1  part 0 | second stream: optimizer over the fusion stack's / classifier's tensors | part 1 + optimizer over the encoders'
  UB  part 0 | second stream: optimizer over EVERYTHING | part 1       (numerically meaningless: an upper bound on what can hide)

prints ms per step for each."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mer_amd import runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402
from mer_amd.optim import FusedAdam  # noqa: E402


def main():
    runtime.require_gpu()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    cfg, B, L = wl["cfg"], wl["B"], wl["L"]
    torch.manual_seed(0)
    model = M2FNet(cfg, precision="bf16", shape_buckets=False).to(dev).train()
    opt = FusedAdam(model, lr=5e-5, weight_decay=0.01)
    text, audio, mask, emotion = bench.synthetic_batch(cfg, B, L, 0, dev)
    eng = model.engine()
    plan = eng.plan(B, L, True, True, None)
    main_s = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    hi = torch.cuda.Stream(device=dev, priority=-1)

    def masked_stream(pattern):
        """a stream whose kernels may only use the compute units of `pattern` (8 x 32 bits; hipExtStreamCreateWithCUMask)"""
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        st = ctypes.c_void_p()
        words = (ctypes.c_uint32 * 8)(*([pattern] * 8))
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
        return torch.cuda.ExternalStream(st.value, device=dev)
    split = int(plan.split_offset())
    n = eng.flat.numel()
    out = {"split": split, "n": n}

    def timed(fn, stream):
        with torch.cuda.stream(stream):
            plan.set_inputs(text, audio, mask, emotion)
            eng.publish_grads()
            for _ in range(5):
                fn(stream)
            torch.cuda.synchronize()
            res = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(steps):
                    fn(stream)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / steps * 1e3)
        return sorted(res)[1]

    def S(st):
        plan.params_fresh(eng.shadows_fresh())
        plan.step(0.1, False, False, True)
        opt.step()

    def P(st):
        plan.params_fresh(eng.shadows_fresh())
        plan.step_part(0, 0.1, False, False, True)
        plan.step_part(1, 0.1, False, False, True)
        opt.step()

    def overlapped(whole):
        def f(st):
            plan.params_fresh(eng.shadows_fresh())
            plan.step_part(0, 0.1, False, False, True)
            side.wait_stream(st)
            plan.step_part(1, 0.1, False, False, True)
            if whole:
                with torch.cuda.stream(side):
                    opt.step()
            else:
                # step_ranges launches in order; the first range goes to the second stream, the second stays here
                def before(i):
                    pass
                opt._step += 1
                g = opt.param_groups[0]
                with torch.cuda.stream(side):
                    runtime.adam_step_shadowed(eng.cfg, eng.flat, eng.ensure_grad(), opt._m, opt._v, eng.wshadow, opt._step, g["lr"], g["betas"],
                                               g["eps"], g["weight_decay"], None, first=split, end=-1)
                runtime.adam_step_shadowed(eng.cfg, eng.flat, eng.ensure_grad(), opt._m, opt._v, eng.wshadow, opt._step, g["lr"], g["betas"],
                                           g["eps"], g["weight_decay"], None, first=0, end=split)
                eng.mark_shadows_fresh()
            st.wait_stream(side)
        return f

    opt._bind()
    out["S_whole_step_then_optimizer_ms"] = timed(S, main_s)
    if split > 0:
        out["P_two_parts_then_optimizer_ms"] = timed(P, main_s)
        out["O1_tail_optimizer_under_part1_ms"] = timed(overlapped(False), main_s)
        out["UB_whole_optimizer_under_part1_ms"] = timed(overlapped(True), main_s)
        out["O1_high_priority_chain_ms"] = timed(overlapped(False), hi)
        out["UB_high_priority_chain_ms"] = timed(overlapped(True), hi)
        def adam_plain(st):
            opt.step()
        out["adam_alone_ms"] = timed(adam_plain, main_s)
        for name, pat in (("1_of_4", 0x11111111), ("1_of_2", 0x55555555), ("low_half_words", 0x0000FFFF), ("1_of_8", 0x01010101)):
            try:
                side = masked_stream(pat)
            except Exception as e:                     # noqa: BLE001
                out["mask_" + name] = str(e)
                continue

            def adam_only(st):
                with torch.cuda.stream(side):
                    opt.step()
                st.wait_stream(side)
            out[f"adam_alone_on_mask_{name}_ms"] = timed(adam_only, main_s)
            out[f"O1_mask_{name}_ms"] = timed(overlapped(False), main_s)
            out[f"UB_mask_{name}_ms"] = timed(overlapped(True), main_s)
    out["S_again_ms"] = timed(S, main_s)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
