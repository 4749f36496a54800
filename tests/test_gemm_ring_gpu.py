"""The RING form of the bf16 GEMM (csrc/gemm.hip, m2f_gemm16_ring_kernel) against the 64x64 build it replaces on launches
that fill the chip.

Both forms add the products of one output element in the same order (k ascending, one MFMA accumulation chain) and apply the
same epilogue operations per element, so the results must be IDENTICAL BIT FOR BIT - for every epilogue term (bias, ReLU,
residual, gate, accumulate, dropout), for two-segment reductions (the fusion layer's cat(x, text) operand), for ragged edges
(rows / columns / reduction lengths that are no multiple of the tile) and for outputs whose leading dimension rules out
16-byte stores.  `tile=64` pins the register-staged 64x64 build; `tile=0` (automatic) takes the ring form: 128x128 tiles from
200 such tiles on, 128x64 tiles from 150, 64x64 tiles from 80 (csrc/gemm.hip, launch_tile16) - asserted through
m2f_gemm_ring_launches().  The register-staged build itself is pinned to fp32 torch references by tests/test_kernels_gpu.py.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ring_count():
    from mer_amd import runtime
    return runtime.lib().m2f_gemm_ring_launches()


def _both(**kw):
    """-> (64x64 build, ring form) of the same call; accumulate cases get identical pre-filled outputs."""
    from mer_amd import functional as F, runtime
    out0 = kw.pop("out0", None)
    outs = []
    expect_ring = kw.pop("expect_ring", True)
    for tile in (64, 0):
        out = out0.clone() if out0 is not None else None
        before = _ring_count()
        c = F.gemm(precision=runtime.BF16, layout=F.NT, src16=True, tile=tile, out=out, **kw)
        torch.cuda.synchronize()
        took_ring = _ring_count() - before
        assert took_ring == (1 if tile == 0 and expect_ring else 0), f"tile={tile}: ring launches {took_ring}"
        outs.append(c)
    return outs


# N per tile configuration at M ~ 1024: 8 x 26 = 208 tiles of 128x128 | 8 x 21 = 168 of 128x64 (88 of 128x128) | 16 x 8 = 128 of
# 64x64 (64 of 128x64).  The 256x128 form (M = 4096) compiles only {-, residual} x {-, GELU} epilogues: the other terms make the
# launcher fall back to 128x128 ring tiles, GELU on the smaller forms to the register-staged build
WIDTH = {"128x128": 3328, "128x64": 1344, "64x64": 512, "256x128": 4096}
ROWS = {"256x128": 4096}                 # 16 x 32 = 512 tiles of 256x128: the text-encoder-sized launches


@pytest.mark.skipif(os.environ.get("M2F_RING", "1") == "0", reason="ring form switched off in the environment")
@pytest.mark.parametrize("form", sorted(WIDTH))
@pytest.mark.parametrize("case", ["plain", "ragged_scalar_stores", "ragged_vector_stores", "bias_relu", "residual_dropout",
                                  "gate_accumulate", "all_terms", "two_segments_relu_a", "short_k", "gelu", "gelu_residual"])
def test_ring_form_equals_64x64_build_bit_for_bit(case, form):
    g = torch.Generator(device=DEV).manual_seed(sum(map(ord, case + form)))
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    M, N, K = ROWS.get(form, 1024), WIDTH[form], 1024
    kw = {}
    if case == "ragged_scalar_stores":
        M, N, K = 1001, N + 2, 200                  # ldc % 4 != 0: no 16-byte stores anywhere; k tail of 8 (one chunk)
    elif case == "ragged_vector_stores":
        M, N, K = 1001, N + 8, 1000                 # interior tiles vectorised, edge tiles element-wise; k tail of 40
    elif case == "short_k":
        K = 40                                      # fewer k-tiles than ring slots
    a, b = rn(M, K), rn(N, K)
    if case in ("bias_relu", "all_terms", "ragged_scalar_stores", "ragged_vector_stores"):
        kw.update(bias=rn(N), relu_out=True)
    if case in ("residual_dropout", "all_terms"):
        kw.update(res=rn(M, N), drop_site=7, drop_p=0.4, rng=torch.tensor([11, 22, 3, 0], dtype=torch.int32, device=DEV))
    if case in ("gate_accumulate", "all_terms"):
        kw.update(gate=rn(M, N), gate_scale=1.25, accumulate=True, out0=rn(M, N))
    if case in ("gelu", "gelu_residual"):
        kw.update(bias=rn(N), relu_out=2, expect_ring=form == "256x128")       # relu_out = 2: exact-erf GELU epilogue
        if case == "gelu_residual":
            kw.update(res=rn(M, N))
    if case == "two_segments_relu_a":
        kw.update(a1=rn(M, 768), b1=rn(N, 768), relu_a=True, bias=rn(N))
    ref, new = _both(a=a, b=b, **kw)
    assert torch.isfinite(ref).all()
    diff = ref != new
    assert not diff.any(), f"{int(diff.sum())} of {ref.numel()} elements differ, max {(ref - new).abs().max().item()}"


@pytest.mark.skipif(os.environ.get("M2F_RING", "1") == "0", reason="ring form switched off in the environment")
def test_ring_form_is_not_taken_below_the_threshold_or_for_pinned_tiles():
    from mer_amd import functional as F, runtime
    a, b = torch.randn(256, 256, device=DEV), torch.randn(448, 256, device=DEV)
    before = _ring_count()
    F.gemm(a, b, F.NT, runtime.BF16, src16=True)              # 28 tiles of 64x64: the register-staged 64x64 build
    F.gemm(a, b, F.NT, runtime.F32)                            # fp32 mode never
    torch.cuda.synchronize()
    assert _ring_count() == before
