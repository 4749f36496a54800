#!/usr/bin/env python
"""Kernel-level check + measurement of the eight-phase 256x256 bf16 GEMM (csrc/gemm_p8.h) through the C ABI (m2f_gemm_p8).

  python tools/p8_bench.py check          # correctness against fp32 products of the same bf16 operands (asymmetric random data)
  python tools/p8_bench.py bench [reps]   # TFLOP/s per shape, random operands, torch events around `reps` launches
  python tools/p8_bench.py race [rounds]  # the same launch over and over on several shapes: every result must equal the first, bit for bit
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mer_amd  # noqa: E402,F401
from mer_amd import runtime  # noqa: E402

dev = torch.device("cuda:0")
lib = runtime.lib()
SCRATCH = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16)


def run(rc, M, N, K, a16, b16, c, bias=None, res=None, act=0, relu_a=0, relu_b=0, bias_grad=None, n_wg=0, prepared=False):
    lda = a16.shape[1]
    ldb = b16.shape[1]
    r = lib.m2f_gemm_p8(rc, M, N, K, ptr(a16), lda, ptr(b16), ldb, ptr(c), c.shape[1], ptr(bias), ptr(res), res.shape[1] if res is not None else 0,
                        act, relu_a, relu_b, ptr(bias_grad), ptr(SCRATCH), -SCRATCH.numel() if prepared else SCRATCH.numel(), n_wg, None)
    if r != 0:
        raise RuntimeError(f"m2f_gemm_p8 -> {r}")


def operands(rc, M, N, K, seed, pad=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if rc:                                            # (leading dimensions are multiples of 8: 16-byte rows, as every bf16 shadow has)
        a = torch.randn(K, (M + 7) // 8 * 8 + pad, generator=g) * 0.5
        b = torch.randn(K, (N + 7) // 8 * 8 + pad, generator=g) * 0.5 + 0.1
    else:
        a = torch.randn(M, K + pad, generator=g) * 0.5
        b = torch.randn(N, K + pad, generator=g) * 0.5 + 0.1
    return a.to(dev).to(torch.bfloat16), b.to(dev).to(torch.bfloat16)


def check():
    worst = 0.0
    cases = [(1, 256, 256, 64), (1, 256, 256, 128), (1, 512, 768, 1024), (1, 300, 768, 1024), (1, 768, 300, 1024), (1, 7, 768, 1024),
             (1, 768, 768, 48), (1, 1024, 2048, 1000), (1, 2304, 768, 1024),
             (0, 256, 256, 64), (0, 512, 768, 1024), (0, 1000, 520, 192), (0, 4096, 1024, 1024), (0, 260, 252, 64)]
    for rc, M, N, K in cases:
        for variant in range(3 if rc else 4):
            a, b = operands(rc, M, N, K, 7 + variant, pad=8 * (variant & 1))
            c = torch.full((M, N + 4 * (variant & 1)), float("nan"), device=dev)
            kw = {}
            if rc:
                relu_a, relu_b = False, variant == 2           # (ReLU on A is not this form's: the launchers refuse it)
                bg = torch.full((M,), float("nan"), device=dev) if variant != 1 else None
                run(1, M, N, K, a.view(torch.int16), b.view(torch.int16), c, relu_a=int(relu_a), relu_b=int(relu_b), bias_grad=bg)
                af, bfl = a.float()[:, :M], b.float()[:, :N]
                if bg is not None:
                    e = (bg - af.sum(0)).abs().max().item() / max(af.sum(0).abs().max().item(), 1e-6)
                    assert e < 2e-3, ("bias_grad", rc, M, N, K, variant, e)
                if relu_a:
                    af = af.clamp_min(0)
                if relu_b:
                    bfl = bfl.clamp_min(0)
                ref = af.double().t() @ bfl.double()
            else:
                bias = torch.randn(N, device=dev) if variant >= 1 else None
                res = torch.randn(M, N + 4, device=dev) if variant >= 2 else None
                act = [0, 1, 2, 0][variant]
                run(0, M, N, K, a.view(torch.int16), b.view(torch.int16), c, bias=bias, res=res, act=act)
                ref = a.float()[:, :K].double() @ b.float()[:, :K].double().t()
                if bias is not None:
                    ref = ref + bias.double()
                if act == 1:
                    ref = ref.clamp_min(0)
                if act == 2:
                    ref = torch.nn.functional.gelu(ref)
                if res is not None:
                    ref = ref + res[:, :N].double()
            torch.cuda.synchronize()
            got = c[:, :N].double()
            assert torch.isfinite(got).all(), ("non-finite result", rc, M, N, K, variant)
            if c.shape[1] > N:
                assert torch.isnan(c[:, N:]).all(), ("wrote past the row", rc, M, N, K, variant)
            err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            tol = 2e-3 if (rc or variant != 2) else 4e-3          # (GELU: polynomial erf, 4.3e-4 absolute)
            assert err < tol, (rc, M, N, K, variant, err)
            worst = max(worst, err)
    print(f"p8 check ok: {len(cases)} shapes, worst relative error {worst:.2e}")


def bench(reps=20):
    out = []
    shapes = [(1, 4096, 4096, 1024), (1, 8192, 8192, 1024), (1, 4096, 4096, 4096), (1, 16384, 8192, 1024), (1, 2048, 1024, 1024), (1, 3072, 1024, 1024),
              (0, 4096, 4096, 4096), (0, 8192, 8192, 8192), (0, 32768, 1024, 1024), (0, 32768, 4096, 1024), (0, 32768, 1024, 4096), (0, 32768, 3072, 1024)]
    for rc, M, N, K in shapes:
        a, b = operands(rc, M, N, K, 3)
        c = torch.empty(M, N, device=dev)
        for _ in range(3):
            run(rc, M, N, K, a.view(torch.int16), b.view(torch.int16), c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(rc, M, N, K, a.view(torch.int16), b.view(torch.int16), c, prepared=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        out.append({"form": "RC (A^T B)" if rc else "KC (A B^T)", "M": M, "N": N, "K": K, "us": ms * 1e3, "TFLOP/s": tf, "frac_of_2.5PF": tf / 2500.0})
        print(json.dumps(out[-1]))
    return out


def race(rounds=30):
    for rc, M, N, K in [(1, 1024, 768, 1024), (1, 2048, 1024, 1024), (0, 2048, 1024, 1024), (1, 768, 768, 192), (0, 4096, 4096, 256)]:
        a, b = operands(rc, M, N, K, 11)
        first = None
        for r in range(rounds):
            c = torch.full((M, N), float("nan"), device=dev)
            run(rc, M, N, K, a.view(torch.int16), b.view(torch.int16), c, n_wg=0 if r % 2 == 0 else 7)
            torch.cuda.synchronize()
            if first is None:
                first = c.clone()
            else:
                assert torch.equal(c, first), ("run-to-run difference", rc, M, N, K, r, (c - first).abs().max().item())
    print(f"p8 race ok: {rounds} rounds per shape, every result bit-identical to the first")


if __name__ == "__main__":
    runtime.require_gpu()
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        check()
    elif what == "bench":
        bench(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif what == "race":
        race(int(sys.argv[2]) if len(sys.argv) > 2 else 30)
