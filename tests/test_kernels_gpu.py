"""GPU parity tests of each HIP kernel (called through the C ABI) against plain fp32 tensor math.

Tolerances: fp32 mode (exact-fp32 MFMA) 2e-5 relative to the output scale; bf16 mode (operands rounded
to bf16, fp32 accumulate) 1.5e-2 relative to the output scale.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import mer_amd  # noqa: E402
from mer_amd import functional as F  # noqa: E402
from mer_amd import runtime  # noqa: E402
from oracle import m2fnet_oracle as O  # noqa: E402

DEV = "cuda"
TOL = {runtime.F32: 2e-5, runtime.BF16: 1.5e-2}


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _close(a, b, tol, what=""):
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 1e-7, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


@pytest.mark.parametrize("prec", [runtime.F32, runtime.BF16])
@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(70, 50, 45), (512, 768, 768), (129, 300, 300), (64, 7, 768), (33, 2048, 96)])
def test_gemm_layouts(prec, tile, shape):
    M, N, K = shape
    a = _rand(M, K, seed=1)
    b_nk = _rand(N, K, seed=2)
    ref = a.double() @ b_nk.double().t()
    _close(F.gemm(a, b_nk, F.NT, prec, tile=tile).double(), ref, TOL[prec], "NT")
    b_kn = b_nk.t().contiguous()
    _close(F.gemm(a, b_kn, F.NN, prec, tile=tile).double(), ref, TOL[prec], "NN")
    a_km = a.t().contiguous()
    c, bg = F.gemm(a_km, b_kn, F.TN, prec, tile=tile, bias_grad=True)
    _close(c.double(), ref, TOL[prec], "TN")
    _close(bg.double(), a_km.double().sum(dim=0), 1e-5, "bias_grad")


@pytest.mark.parametrize("prec", [runtime.F32, runtime.BF16])
@pytest.mark.parametrize("shape", [(512, 768, 768), (512, 7, 768), (100, 130, 2048), (64, 64, 512)])
def test_gemm_split_k_matches_unsplit_and_is_reproducible(prec, shape):
    """Small launches are split along K inside the launch (last-arriver reduce); results must equal the unsplit
    kernel up to summation order, be bitwise reproducible run to run, and keep the fused epilogue."""
    M, N, K = shape
    a, b, bias, res = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3), _rand(M, N, seed=4)
    ref = torch.relu(a.double() @ b.double().t() + bias.double()) + res.double()
    outs = [F.gemm(a, b, F.NT, prec, bias=bias, res=res, relu_out=True, tile=64, split_k=True) for _ in range(4)]
    _close(outs[0].double(), ref, TOL[prec], "split-K NT")
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "split-K must be bitwise reproducible (fixed-order reduce)"
    plain = F.gemm(a, b, F.NT, prec, bias=bias, res=res, relu_out=True, tile=64, split_k=False)
    _close(outs[0].double(), plain.double(), 1e-5 if prec == runtime.F32 else 1e-5, "split vs unsplit")
    bt = b.t().contiguous()
    _close(F.gemm(a, bt, F.NN, prec, tile=64, split_k=True).double(), a.double() @ b.double().t(), TOL[prec], "split-K NN")
    # two-segment operand under split-K
    a2, b2 = _rand(M, 256, seed=5), _rand(N, 256, seed=6)
    got = F.gemm(a, b, F.NT, prec, a1=a2, b1=b2, tile=64, split_k=True)
    _close(got.double(), a.double() @ b.double().t() + a2.double() @ b2.double().t(), TOL[prec], "split-K two segments")


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(70, 50, 45), (512, 768, 768), (129, 300, 300), (512, 900, 300), (64, 7, 768), (33, 2048, 96),
                                   (512, 768, 2048), (200, 304, 520)])
def test_gemm_bf16_source_equals_fp32_source(tile, shape):
    """bf16 mode staged from the bf16 shadows must give the SAME result as staging from fp32 (same rounding points)."""
    M, N, K = shape
    prec = runtime.BF16
    a, b_nk = _rand(M, K, seed=1), _rand(N, K, seed=2)
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    ref = F.gemm(a, b_nk, F.NT, prec, bias=bias, res=res, relu_out=True, tile=tile)
    got = F.gemm(a, b_nk, F.NT, prec, bias=bias, res=res, relu_out=True, tile=tile, src16=True)
    _close(got, ref, 2e-6, "NT bf16-source")
    b_kn = b_nk.t().contiguous()
    _close(F.gemm(a, b_kn, F.NN, prec, tile=tile, src16=True), F.gemm(a, b_kn, F.NN, prec, tile=tile), 2e-6, "NN bf16-source")
    a_km = a.t().contiguous()
    c1, bg1 = F.gemm(a_km, b_kn, F.TN, prec, tile=tile, bias_grad=True, src16=True)
    c0, bg0 = F.gemm(a_km, b_kn, F.TN, prec, tile=tile, bias_grad=True)
    _close(c1, c0, 2e-6, "TN bf16-source")
    _close(bg1, a_km.to(torch.bfloat16).float().sum(dim=0), 1e-5, "bias grad from bf16 operands")
    # relu prologue + two segments
    a2, b2 = _rand(M, 64, seed=5), _rand(N, 64, seed=6)
    r0 = F.gemm(a, b_nk, F.NT, prec, a1=a2, b1=b2, relu_a=True, tile=tile)
    r1 = F.gemm(a, b_nk, F.NT, prec, a1=a2, b1=b2, relu_a=True, tile=tile, src16=True)
    _close(r1, r0, 2e-6, "two segments + relu prologue")


def test_gemm_large_m_tile_and_gelu_epilogue():
    """The text encoder's GEMMs: M = utterances x tokens is large enough for the 256x128 tiles (auto-selected at >= 1024 of
    them), bias + exact-erf GELU epilogue; ragged edges in both dimensions."""
    M, N, K = 16384 + 37, 2048 + 24, 264
    a, b = _rand(M, K, seed=21), _rand(N, K, seed=22)
    bias = _rand(N, seed=23)
    got = F.gemm(a, b, F.NT, runtime.BF16, bias=bias, relu_out=2, src16=True)
    pre = a.to(torch.bfloat16).float() @ b.to(torch.bfloat16).float().t() + bias
    ref = 0.5 * pre * (1.0 + torch.erf(pre / 2.0 ** 0.5))
    _close(got, ref, 2e-4, "large-M bf16 GEMM + GELU")
    got32 = F.gemm(a[:300], b[:200], F.NT, runtime.F32, bias=bias[:200], relu_out=2)
    pre32 = a[:300] @ b[:200].t() + bias[:200]
    _close(got32, 0.5 * pre32 * (1.0 + torch.erf(pre32 / 2.0 ** 0.5)), 1e-5, "fp32 GEMM + GELU")


# (shapes 3 and 4 take the ring form, gemm_ring_256x128_fp8.hip: edge tiles + a k remainder, and whole tiles incl. the e4m3 result; the last three the
#  eight-phase form, gemm_p8.h EPI 4 on v_mfma_scale_f32_16x16x128_f8f6f4: 256 tiles of 256x256 = one round, whole tiles, an edge row of tiles, a long k)
@pytest.mark.parametrize("shape", [(300, 200, 64), (1024, 768, 768), (4096 * 9 + 5, 1024 + 40, 256 + 48), (8192, 1024, 512),
                                   (4096, 4096, 256), (4000, 4096, 384), (16384, 1024, 1024)])
def test_gemm_fp8_matches_dequantised_reference(shape):
    """fp8 (OCP e4m3) operands, fp32 accumulate: exact against an fp32 product of the SAME quantised values (the MFMA
    multiplies e4m3 x e4m3 exactly and accumulates in fp32), with the de-quantisation scale, bias, GELU and residual."""
    M, N, K = shape
    a, b = _rand(M, K, seed=41), _rand(N, K, seed=42) * 0.05
    bias, res = _rand(N, seed=43), _rand(M, N, seed=44)
    sa, sb = 16.0, 448.0 / b.abs().max().item()
    a8, b8 = (a * sa).to(torch.float8_e4m3fn), (b * sb).to(torch.float8_e4m3fn)
    pre = (a8.float() @ b8.float().t()) / (sa * sb) + bias
    got = F.gemm_fp8(a8, b8, 1.0 / (sa * sb), bias=bias, res=res, activation=2)
    ref = 0.5 * pre * (1.0 + torch.erf(pre / 2.0 ** 0.5)) + res
    _close(got, ref, 1e-3, "fp8 GEMM + GELU + residual")            # accumulation order + the 4.3e-4 polynomial erf
    plain = (a8.float() @ b8.float().t()) / (sa * sb)
    _close(F.gemm_fp8(a8, b8, 1.0 / (sa * sb)), plain, 1e-4, "fp8 GEMM plain")
    # e4m3 output (the FFN hidden layer never exists in fp32): same values as quantising the fp32 result
    out8 = torch.empty(M, N, dtype=torch.float8_e4m3fn, device=DEV)
    F.gemm_fp8(a8, b8, 1.0 / (sa * sb), bias=bias, activation=2, out8=out8, out8_scale=8.0)
    want8 = F.quantize_fp8((0.5 * pre * (1.0 + torch.erf(pre / 2.0 ** 0.5))).contiguous(), 8.0)
    diff = (out8.float() - want8.float()).abs()
    assert (diff > 0).float().mean().item() < 0.10             # the kernel's erf is a 4.3e-4 polynomial: a few % of the values land on the other side of an e4m3 rounding boundary
    # ... and then by one e4m3 step (2^-9 in the subnormal range) plus the polynomial's absolute error (<= 0.5 |x| 4.3e-4,
    # times the output scale 8): e.g. GELU of x < -4 is exactly 0 in the kernel and -1e-4 in the reference
    assert (diff <= 0.13 * want8.float().abs() + 2.0 ** -9 + 0.015).all()


def test_quantize_fp8_matches_torch_and_saturates():
    x = _rand(1000, 64, seed=51) * 100.0
    got = F.quantize_fp8(x, 2.0)
    ref = (x * 2.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    assert torch.equal(got.float(), ref.float())
    assert got.float().abs().max().item() == 448.0


@pytest.mark.parametrize("prec", [runtime.F32, runtime.BF16])
def test_gemm_epilogue_and_segments(prec):
    M, N, K0, K1 = 96, 80, 64, 40
    a0, a1 = _rand(M, K0, seed=3), _rand(M, K1, seed=4)
    w = _rand(N, K0 + K1, seed=5)
    bias, res, gate = _rand(N, seed=6), _rand(M, N, seed=7), _rand(M, N, seed=8)
    pre = torch.relu(torch.cat((a0, a1), dim=1)).double() @ w.double().t() + bias.double()
    ref = (torch.relu(pre) + res.double()) * (gate > 0).double() * 1.5
    out = F.gemm(a0, w[:, :K0], F.NT, prec, a1=a1, b1=w[:, K0:], bias=bias, res=res, gate=gate, gate_scale=1.5,
                 relu_a=True, relu_out=True)
    _close(out.double(), ref, TOL[prec], "fused epilogue")
    base = _rand(M, N, seed=9)
    acc = base.clone()
    F.gemm(a0, w[:, :K0], F.NT, prec, out=acc, accumulate=True)
    _close(acc.double(), base.double() + a0.double() @ w[:, :K0].double().t(), TOL[prec], "accumulate")
    # strided C (column block of a wider matrix) + relu on the B operand of a wgrad
    big = torch.zeros(N, 2 * K0, device=DEV)
    dy = _rand(M, N, seed=10)
    F.gemm(dy, a0, F.TN, prec, out=big[:, K0:], relu_b=True)
    _close(big[:, K0:].double(), dy.double().t() @ torch.relu(a0).double(), TOL[prec], "wgrad strided")
    assert big[:, :K0].abs().max().item() == 0.0


def test_gemm_dropout_epilogue_statistics_and_replay():
    M, N, K, p = 256, 512, 64, 0.4
    a, b = _rand(M, K, seed=11), _rand(N, K, seed=12)
    rng = torch.tensor([123, 456, 7, 0], dtype=torch.int32, device=DEV)
    plain = F.gemm(a, b)
    d1 = F.gemm(a, b, drop_site=5, drop_p=p, rng=rng)
    d2 = F.gemm(a, b, drop_site=5, drop_p=p, rng=rng)
    assert torch.equal(d1, d2), "mask must be a pure function of (state, site, index)"
    kept = d1 != 0
    rate = kept.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    _close(d1[kept].double(), (plain[kept] / (1 - p)).double(), 1e-6, "kept values scaled by 1/(1-p)")
    d3 = F.gemm(a, b, drop_site=6, drop_p=p, rng=rng)
    assert (d3 != 0).ne(kept).float().mean().item() > 0.3, "different sites must give different masks"
    rng2 = rng.clone()
    rng2[2] += 1
    d4 = F.gemm(a, b, drop_site=5, drop_p=p, rng=rng2)
    assert (d4 != 0).ne(kept).float().mean().item() > 0.3, "different steps must give different masks"
    # rows / columns are not correlated
    assert abs(kept.float().mean(dim=0).std().item()) < 0.06 and abs(kept.float().mean(dim=1).std().item()) < 0.06


def _attn_case(B, L, H, hd, seed, lengths=None):
    E = H * hd
    q, k, v = _rand(B * L, E, seed=seed), _rand(B * L, E, seed=seed + 1), _rand(B * L, E, seed=seed + 2)
    key_pad = torch.zeros(B, L, dtype=torch.bool)
    if lengths:
        for b, n in enumerate(lengths):
            key_pad[b, n:] = True
    return q, k, v, key_pad.to(DEV)


@pytest.mark.parametrize("B,L,H,hd,lengths", [
    (4, 16, 8, 96, [16, 9, 1, 12]), (3, 33, 4, 15, [33, 17, 2]), (2, 9, 2, 24, None), (2, 64, 2, 128, [64, 40]),
    (5, 7, 3, 75, [7, 7, 3, 1, 5])])
def test_attention_forward_backward(B, L, H, hd, lengths):
    q, k, v, key_pad = _attn_case(B, L, H, hd, 20, lengths)
    out, probs = F.attention_fwd(q, k, v, key_pad, B, L, H)
    E = H * hd
    qr, kr, vr = (t.detach().clone().view(B, L, E).requires_grad_(True) for t in (q, k, v))
    ref, p_ref = O.attention(qr, kr, vr, key_pad, H, return_probs=True)
    _close(out.view(B, L, E), ref.detach(), 2e-5, "attention out")
    Lp = probs.shape[-1]
    _close(probs.view(B, H, Lp, Lp)[:, :, :L, :L].transpose(-1, -2), p_ref.detach(), 2e-5, "probs")
    dout = _rand(B * L, E, seed=30)
    ref.backward(dout.view(B, L, E))
    dq, dk, dv = F.attention_bwd(q, k, v, key_pad, out, probs, dout, B, L, H)
    _close(dq.view(B, L, E), qr.grad, 3e-5, "dq")
    _close(dk.view(B, L, E), kr.grad, 3e-5, "dk")
    _close(dv.view(B, L, E), vr.grad, 3e-5, "dv")


def test_attention_cross_strided_operands():
    """FusionAttentionModule form: q and v live in one [T, 2E] buffer, k in another (src/model.py:14)."""
    B, L, H, hd = 3, 12, 4, 32
    E = H * hd
    qv = _rand(B * L, 2 * E, seed=40)
    k = _rand(B * L, E, seed=41)
    key_pad = torch.zeros(B, L, dtype=torch.bool, device=DEV)
    key_pad[1, 5:] = True
    out, probs = F.attention_fwd(qv[:, :E], k, qv[:, E:], key_pad, B, L, H)
    ref = O.attention(qv[:, :E].reshape(B, L, E), k.view(B, L, E), qv[:, E:].reshape(B, L, E), key_pad, H)
    _close(out.view(B, L, E), ref, 2e-5, "cross attention")


def test_attention_dropout_backward_matches_autograd_with_same_mask():
    B, L, H, hd, p = 2, 16, 2, 32, 0.4
    E = H * hd
    q, k, _, key_pad = _attn_case(B, L, H, hd, 50, [16, 11])
    # V = [I_L | 0] per head exposes the dropped probabilities in the output
    v = torch.zeros(B, L, H, hd, device=DEV)
    for j in range(L):
        v[:, j, :, j] = 1.0
    v = v.view(B * L, E)
    rng = torch.tensor([9, 8, 3, 0], dtype=torch.int32, device=DEV)
    out, probs = F.attention_fwd(q, k, v, key_pad, B, L, H, drop_site=3, drop_p=p, rng=rng)
    Lp = probs.shape[-1]
    P = probs.view(B, H, Lp, Lp)[:, :, :L, :L].transpose(-1, -2)             # [B,H,i,j] pre-dropout
    Pd = out.view(B, L, H, hd)[..., :L].permute(0, 2, 1, 3)                   # dropped probabilities
    mask = (Pd != 0)
    valid = P > 1e-12
    rate = mask[valid].float().mean().item()
    assert abs(rate - (1 - p)) < 0.08, rate
    _close(Pd[mask], (P / (1 - p))[mask], 1e-5, "dropped probs scaled")
    # autograd reference with the extracted mask and a random V
    v2 = _rand(B * L, E, seed=55)
    out2, probs2 = F.attention_fwd(q, k, v2, key_pad, B, L, H, drop_site=3, drop_p=p, rng=rng)
    qr, kr, vr = (t.detach().clone().view(B, L, H, hd).permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v2))
    s = (qr @ kr.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, dim=-1) * mask.float() / (1 - p)
    o_ref = (pr @ vr).permute(0, 2, 1, 3).reshape(B * L, E)
    _close(out2, o_ref.detach(), 2e-5, "dropout forward")
    dout = _rand(B * L, E, seed=56)
    o_ref.backward(dout)
    dq, dk, dv = F.attention_bwd(q, k, v2, key_pad, out2, probs2, dout, B, L, H, drop_site=3, drop_p=p, rng=rng)
    back = lambda g: g.permute(0, 2, 1, 3).reshape(B * L, E)
    _close(dq, back(qr.grad), 3e-5, "dq dropout")
    _close(dk, back(kr.grad), 3e-5, "dk dropout")
    _close(dv, back(vr.grad), 3e-5, "dv dropout")


@pytest.mark.parametrize("T,d", [(50, 768), (512, 300), (33, 50), (17, 2048), (64, 1024)])
def test_layernorm_forward_backward(T, d):
    x, g, b, res = _rand(T, d, seed=60, scale=3.0), 1 + 0.1 * _rand(d, seed=61), 0.1 * _rand(d, seed=62), _rand(T, d, seed=63)
    out, stats = F.layernorm_fwd(x, g, b, res)
    xr, gr, br = (t.detach().clone().requires_grad_(True) for t in (x, g, b))
    ref = res + O.layer_norm(xr, gr, br)
    _close(out, ref.detach(), 1e-5, "ln fwd")
    dy, extra = _rand(T, d, seed=64), _rand(T, d, seed=65)
    ref.backward(dy)
    dx, dg, db = F.layernorm_bwd(x, g, stats, dy, extra)
    _close(dx, xr.grad + extra, 2e-5, "ln dx")
    _close(dg, gr.grad, 2e-5, "ln dgamma")
    _close(db, br.grad, 2e-5, "ln dbeta")


@pytest.mark.parametrize("weighted", [False, True])
def test_cross_entropy_matches_torch(weighted):
    T, C = 300, 7
    logits = _rand(T, C, seed=70, scale=2.0)
    g = torch.Generator().manual_seed(71)
    labels = torch.randint(0, C, (T,), generator=g)
    labels[torch.rand(T, generator=g) < 0.3] = -1
    labels = labels.to(DEV)
    w = torch.tensor([0.3, 1.2, 2.1, 1.3, 1.2, 5.3, 5.2], device=DEV) if weighted else None
    lr = logits.detach().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(weight=w, ignore_index=-1, label_smoothing=0.1)(lr, labels)
    ref.backward()
    out, dl = F.cross_entropy(logits, labels, w, 0.1, True)
    assert abs(out[0].item() - ref.item()) < 2e-6
    _close(dl, lr.grad, 1e-5, "dlogits")
    out2, dl2 = F.cross_entropy(logits, labels, w, 0.1, False)
    _close(dl2 / out2[1], lr.grad, 1e-5, "unnormalised dlogits / den")


def test_adam_matches_torch():
    n = 4096 + 64
    p0, g = _rand(n, seed=80), _rand(n, seed=81)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        gg = g * step
        ref_p.grad = gg.clone()
        opt.step()
        runtime.adam_step(p, gg, m, v, step, 1e-3, (0.9, 0.999), 1e-8, 0.01)
    _close(p, ref_p.detach(), 1e-6, "adam params")
    # device-side gradient scale (data-parallel global denominator)
    p2, m2, v2 = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    den = torch.tensor([4.0], device=DEV)
    runtime.adam_step(p2, g * 4.0, m2, v2, 1, 1e-3, (0.9, 0.999), 1e-8, 0.01, grad_scale=den)
    p3, m3, v3 = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    runtime.adam_step(p3, g, m3, v3, 1, 1e-3, (0.9, 0.999), 1e-8, 0.01)
    _close(p2, p3, 1e-6, "grad_scale")
    # bf16 gradient input (data-parallel bf16 exchange): identical to the fp32 kernel fed the same rounded gradients
    g16 = g.to(torch.bfloat16)
    p4, m4, v4 = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p5, m5, v5 = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        runtime.adam_step(p4, g16, m4, v4, step, 1e-3, (0.9, 0.999), 1e-8, 0.01, grad_scale=den)
        runtime.adam_step(p5, g16.float(), m5, v5, step, 1e-3, (0.9, 0.999), 1e-8, 0.01, grad_scale=den)
    assert torch.equal(p4, p5) and torch.equal(m4, m5) and torch.equal(v4, v5)


@pytest.mark.parametrize("B,L,H,hd,ld", [(32, 16, 5, 60, 904), (16, 16, 8, 96, 2304), (4, 9, 8, 128, 3072), (3, 33, 4, 32, 392)])
def test_attention_bf16_mode_forms(B, L, H, hd, ld, monkeypatch):
    """bf16 mode runs the dialogue attention kernels in two cheaper forms (AttnBatch::bf16_math; M2F_ATTN_BF16_KERNEL selects them
    for the kernel-level entry points): (a) the Q / K / V (backward: + dO, O) slabs are staged from the operands' bf16 shadows -
    bit for bit what the fp32 kernel computes on inputs rounded to bf16; (b) the head-dim contractions Q K^T and dO V^T run on the
    bf16 MFMA - within bf16 rounding of (a)."""
    from mer_amd import runtime
    torch.manual_seed(B + hd)
    d, T = H * hd, B * L
    ws = torch.randn(2 * T, ld, device=DEV) * 0.5                     # rows [0, T): packed q | k | v, rows [T, 2T): dO | O
    sh = ws.to(torch.bfloat16).contiguous()
    wr = sh.float()
    kp = torch.zeros(B, L, dtype=torch.bool, device=DEV)
    kp[1, max(L // 2, 1):] = True
    valid = ~kp.reshape(-1)

    def shadows(on):
        runtime.check(runtime.lib().m2f_set_shadow_map(ws.data_ptr() if on else None, sh.data_ptr() if on else None, ws.numel() if on else 0), "m2f_set_shadow_map")

    def run(src, mask, with_shadows):
        q, k, v, do = src[:T, :d], src[:T, d:2 * d], src[:T, 2 * d:3 * d], src[T:, :d]
        monkeypatch.setenv("M2F_ATTN_BF16_KERNEL", str(mask))
        shadows(with_shadows)
        try:
            out, probs = F.attention_fwd(q, k, v, kp, B, L, H)
            o_in = src[T:, d:2 * d]
            o_in.copy_(out)                                           # O at a place that has a shadow ...
            if with_shadows:
                sh[T:, d:2 * d].copy_(out.to(torch.bfloat16))          # ... which the forward kernel would have written
            dq, dk, dv = F.attention_bwd(q, k, v, kp, o_in, probs, do, B, L, H)
        finally:
            shadows(False)
            monkeypatch.setenv("M2F_ATTN_BF16_KERNEL", "0")
        return out, dq, dk, dv

    exact = run(ws.clone(), 0, False)                                  # fp32 kernel, unrounded inputs
    rounded_src = wr.clone()
    want = run(rounded_src, 0, False)                                  # fp32 kernel on bf16-rounded inputs
    got = run(ws, 62, True)                                            # staged from the shadows, exact contractions
    # (O enters the backward rounded too: `want` fed the fp32 O of its own forward, so compare the forward bit for bit and the
    #  gradients to the rounding of O)
    assert torch.equal(got[0][valid], want[0][valid])
    for a, b in zip(got[1:], want[1:]):
        assert (a - b)[valid].abs().max().item() <= 2e-2 * b[valid].abs().max().item()
    both = run(ws, 63, True)                                           # + bf16 contractions
    for a, b, e in zip(both, got, exact):
        scale = e[valid].abs().max().item()
        assert (a - b)[valid].abs().max().item() <= 2e-2 * scale
        assert (a - e)[valid].abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize("prec", [runtime.F32, runtime.BF16])
@pytest.mark.parametrize("src16", [False, True])
def test_skinny_classifier_gemms(prec, src16):
    """csrc/skinny.hip behind m2f_gemm: [T, K] x [7, K]^T + bias (the logits) and [T, 7] x [7, N] with the ReLU / dropout gate
    (their input gradient), vectorised and element-wise paths, fp32 and bf16 (from fp32 sources or from bf16 shadows)."""
    if src16 and prec != runtime.BF16:
        pytest.skip("shadows are a bf16-mode thing")
    r = (lambda t: t.to(torch.bfloat16).double()) if prec == runtime.BF16 else (lambda t: t.double())
    for (T, K, ncls) in [(300, 768, 7), (37, 50, 3), (1024, 1024, 8)]:
        h, w, bias = _rand(T, K, seed=61), _rand(ncls, K, seed=62) * 0.1, _rand(ncls, seed=63)
        got = F.gemm(h, w, F.NT, prec, bias=bias, src16=src16)
        _close(got.double(), r(h) @ r(w).t() + bias.double(), 2e-5, f"skinny NT {T}x{ncls}x{K}")       # (the reference rounds the operands as the kernel does)
        dl, gate = _rand(T, ncls, seed=64), _rand(T, K, seed=65)
        got = F.gemm(dl, w, F.NN, prec, gate=gate, gate_scale=1.25, src16=src16)
        ref = (r(dl) @ r(w)) * (gate.double() > 0) * 1.25
        _close(got.double(), ref, 2e-5, f"skinny NN {T}x{K}x{ncls}")
        # the same arithmetic per row whatever the number of rows (packed and padded plans must agree bit for bit)
        assert torch.equal(F.gemm(h[:5].contiguous(), w, F.NT, prec, bias=bias, src16=src16), F.gemm(h, w, F.NT, prec, bias=bias, src16=src16)[:5])
