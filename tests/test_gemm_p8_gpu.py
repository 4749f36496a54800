"""The eight-phase 256x256 bf16 GEMM (csrc/gemm_p8.h: the weight-gradient table launch's kernel since round 4, and the text encoder's
large forward launches) through its kernel-level C entry point m2f_gemm_p8 - the checks of tools/p8_bench.py:
  * both operand forms (row-major RC = dW = dY^T X with bias-gradient row sums and ReLU on the X operand; k-contiguous KC = nn.Linear
    forward with bias / ReLU / GELU / residual) against fp64 products of the SAME bf16 operands, on whole tiles, ragged edges (300- and
    7-wide operands, 252 columns), k tails (48, 1000 rows of reduction), padded leading dimensions, results with a row stride;
  * the same launch repeated, with the tiles dealt to 256 and to 7 workgroups (several tiles per workgroup = the continuous prefetch
    stream across tile boundaries, the relaxed first wait behind an epilogue): every result bit-identical to the first.
A hand-placed LDS-DMA / barrier schedule passes or fails by its counted waits, not by luck: the repeated runs are the race screen the
guide asks for after a change to a synchronisation structure."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _tool():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import p8_bench
    return p8_bench


def test_p8_matches_fp64_products_of_the_same_bf16_operands():
    _tool().check()


def test_p8_results_do_not_depend_on_the_run_or_on_the_tile_distribution():
    _tool().race(12)


def test_relu_on_the_a_operand_is_refused_not_ignored():
    import torch
    pb = _tool()
    a, b = pb.operands(1, 256, 256, 64, 1)
    c = torch.empty(256, 256, device="cuda")
    with pytest.raises(RuntimeError):
        pb.run(1, 256, 256, 64, a.view(torch.int16), b.view(torch.int16), c, relu_a=1)


def test_large_forward_launches_reach_the_eight_phase_kernel_through_the_dispatcher():
    """The production path of the text encoder's large GEMMs: m2f_gemm (bf16 mode, operands with bf16 shadows, automatic tile choice) hands a
    forward-form launch of at least 256 tiles of 256 x 256 to the eight-phase kernel (gemm.hip::launch_tile16; the ring-launch counter moves)
    - bias + GELU + residual, against the fp64 result on the same rounded operands; a launch with a dropout site (not this form's) keeps
    its ring kernel and still agrees with itself run twice."""
    import torch
    import mer_amd  # noqa: F401
    from mer_amd import functional as F, runtime
    g = torch.Generator().manual_seed(5)
    M, N, K = 8192, 2048, 512
    a = (torch.randn(M, K, generator=g) * 0.5).cuda()
    b = (torch.randn(N, K, generator=g) * 0.5 + 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    before = runtime.lib().m2f_gemm_ring_launches()
    c = F.gemm(a, b, layout=F.NT, precision=runtime.BF16, bias=bias, res=res, relu_out=2, src16=True)
    torch.cuda.synchronize()
    assert runtime.lib().m2f_gemm_ring_launches() == before + 1
    ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
    ref = torch.nn.functional.gelu(ar @ br.t() + bias.double()) + res.double()
    err = (c.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 4e-3, err


@pytest.mark.parametrize("M", [8192, 8192 - 100])
def test_bf16_shadow_and_shadow_only_results_of_the_eight_phase_kernel(M):
    """What the text encoder's bf16 data flow relies on (round 4): a result inside the shadow map also leaves as bf16 - sixteen bytes per lane after the
    permlane exchange of gemm_p8.h's epilogue - bit for bit the rounding of the fp32 result; with m2f_set_shadow_only(1) ONLY the bf16 copy is written
    (whole tiles: the fp32 buffer keeps its sentinel; edge tiles still write both).  M = 8,092: the last row of tiles takes the element-wise edge path."""
    import torch
    import mer_amd  # noqa: F401
    from mer_amd import functional as F, runtime
    lib = runtime.lib()
    g = torch.Generator().manual_seed(9)
    N, K = 2048, 512
    a = (torch.randn(M, K, generator=g) * 0.5).cuda()
    b = (torch.randn(N, K, generator=g) * 0.5 + 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ws = torch.full((M * N,), float("nan"), device="cuda")
    sh = torch.zeros(M * N, dtype=torch.bfloat16, device="cuda")
    out = ws.view(M, N)
    runtime.check(lib.m2f_set_shadow_map(ws.data_ptr(), sh.data_ptr(), ws.numel()), "m2f_set_shadow_map")
    try:
        before = lib.m2f_gemm_ring_launches()
        F.gemm(a, b, layout=F.NT, precision=runtime.BF16, bias=bias, relu_out=2, src16=True, out=out)
        torch.cuda.synchronize()
        assert lib.m2f_gemm_ring_launches() == before + 1
        ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
        ref = torch.nn.functional.gelu(ar @ br.t() + bias.double())
        assert (out.double() - ref).abs().max().item() / ref.abs().max().item() < 4e-3
        assert torch.equal(sh.view(M, N), out.to(torch.bfloat16))                 # the shadow IS the rounded fp32 result
        both = sh.clone()
        ws.fill_(float("nan")); sh.zero_()
        runtime.check(lib.m2f_set_shadow_only(1), "m2f_set_shadow_only")
        try:
            F.gemm(a, b, layout=F.NT, precision=runtime.BF16, bias=bias, relu_out=2, src16=True, out=out)
        finally:
            runtime.check(lib.m2f_set_shadow_only(0), "m2f_set_shadow_only")
        torch.cuda.synchronize()
        assert torch.equal(sh, both)                                              # same bits with or without the fp32 store
        whole_rows = M // 256 * 256
        assert torch.isnan(out[:whole_rows]).all()                                # whole tiles: the fp32 buffer was not touched
        if whole_rows < M:
            assert torch.equal(out[whole_rows:].to(torch.bfloat16), sh.view(M, N)[whole_rows:])      # edge tiles keep the fp32 store
    finally:
        runtime.check(lib.m2f_set_shadow_map(None, None, 0), "m2f_set_shadow_map")
