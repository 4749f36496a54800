"""Thin Python wrappers over the kernel-level C entry points (include/m2fnet_hip.h).

Used by the parity tests (each HIP kernel against the oracle) and by the standalone
``FusionAttentionModule.forward``.  Tensors must be fp32 CUDA tensors; strides are passed as leading
dimensions, so column slices of a wider matrix are legal operands.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import runtime
from .runtime import lib, check, ptr, stream_ptr

NT, NN, TN = 0, 1, 2


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, "need fp32 CUDA row-major 2-D"
    return t.stride(0)


_SPLITK = {}


def _splitk_scratch(device):
    """(partial slabs, zeroed tickets, max tiles) per device for the split-K path of small launches."""
    if device not in _SPLITK:
        n = 512
        _SPLITK[device] = (torch.empty(n * 4 * 64 * 64, dtype=torch.float32, device=device),
                           torch.zeros(n, dtype=torch.int32, device=device), n)
    return _SPLITK[device]


def _shadow16(t: torch.Tensor) -> torch.Tensor:
    rows, cols = t.shape
    out = torch.zeros(rows, (cols + 7) // 8 * 8, dtype=torch.bfloat16, device=t.device)
    out[:, :cols] = t.to(torch.bfloat16)
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, layout: int = NT, precision: int = runtime.F32,
         a1: Optional[torch.Tensor] = None, b1: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, gate_scale: float = 1.0,
         bias_grad: bool = False, relu_a: bool = False, relu_b: bool = False, relu_out: bool = False,
         out: Optional[torch.Tensor] = None, accumulate: bool = False, drop_site: int = 0, drop_p: float = 0.0,
         rng: Optional[torch.Tensor] = None, tile: int = 0, split_k: bool = False, src16: bool = False,
         shadows=None):
    """layout NT: a[M,K] b[N,K]; NN: a[M,K] b[K,N]; TN: a[K,M] b[K,N].  Returns C (and bias_grad[M] for TN).
    `shadows` = prebuilt bf16 images (a, a1, b, b1) replacing the per-call ones `src16=True` makes."""
    runtime.require_gpu()
    if layout == NT:
        M, K0 = a.shape; N = b.shape[0]
    elif layout == NN:
        M, K0 = a.shape; N = b.shape[1]
    else:
        K0, M = a.shape; N = b.shape[1]
    K1 = 0
    if a1 is not None:
        K1 = a1.shape[0] if layout == TN else a1.shape[1]
    c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=a.device)
    bg = torch.empty(M, dtype=torch.float32, device=a.device) if bias_grad else None
    ws, tickets, nmax = _splitk_scratch(a.device) if split_k else (None, None, 0)
    # bf16 shadows (test plumbing: in the plan the producer kernels write them): zero-padded to a multiple of 8 columns
    sh = list(shadows) if shadows is not None else [None if (t is None or not src16) else _shadow16(t) for t in (a, a1, b, b1)]
    shp = [(ptr(t), t.stride(0)) if t is not None else (None, 0) for t in sh]
    check(lib().m2f_gemm(precision, layout, M, N, K0, K1, ptr(a), _ld(a), ptr(a1), _ld(a1) if a1 is not None else 0,
                         ptr(b), _ld(b), ptr(b1), _ld(b1) if b1 is not None else 0, ptr(c), _ld(c), ptr(bias),
                         ptr(res), _ld(res) if res is not None else 0, ptr(gate), _ld(gate) if gate is not None else 0,
                         gate_scale, ptr(bg), int(relu_a), int(relu_b), int(relu_out), int(accumulate), drop_site,
                         drop_p, ptr(rng), tile, ptr(ws), ptr(tickets), nmax, shp[0][0], shp[0][1], shp[1][0], shp[1][1],
                         shp[2][0], shp[2][1], shp[3][0], shp[3][1], stream_ptr()), "m2f_gemm")
    return (c, bg) if bias_grad else c


def gemm_fp8(a8: torch.Tensor, b8: torch.Tensor, acc_scale: float, bias: Optional[torch.Tensor] = None,
             res: Optional[torch.Tensor] = None, activation: int = 0, out: Optional[torch.Tensor] = None,
             out8: Optional[torch.Tensor] = None, out8_scale: float = 1.0) -> torch.Tensor:
    """a8 [M, K], b8 [N, K] torch.float8_e4m3fn -> fp32 [M, N] = act(acc_scale * a8 b8^T + bias) + res, or - with `out8` -
    the same result quantised as e4m3(result * out8_scale) into out8 [M, N] (no fp32 output)."""
    runtime.require_gpu()
    assert a8.dtype == torch.float8_e4m3fn and b8.dtype == torch.float8_e4m3fn
    M, K = a8.shape
    N = b8.shape[0]
    if out8 is not None:
        assert out8.dtype == torch.float8_e4m3fn and out8.shape == (M, N)
        check(lib().m2f_gemm_fp8(M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), float(acc_scale), None, out8.stride(0),
                                 ptr(bias), ptr(res), _ld(res) if res is not None else 0, int(activation), ptr(out8),
                                 float(out8_scale), stream_ptr()), "m2f_gemm_fp8")
        return out8
    c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=a8.device)
    check(lib().m2f_gemm_fp8(M, N, K, ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), float(acc_scale), ptr(c), _ld(c), ptr(bias),
                             ptr(res), _ld(res) if res is not None else 0, int(activation), None, 1.0, stream_ptr()), "m2f_gemm_fp8")
    return c


def quantize_fp8(src: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """e4m3(clamp(src * scale, +-448)) of a contiguous fp32 tensor."""
    runtime.require_gpu()
    assert src.is_contiguous() and src.dtype == torch.float32
    dst = out if out is not None else torch.empty(src.shape, dtype=torch.float8_e4m3fn, device=src.device)
    check(lib().m2f_quantize_fp8(ptr(src), ptr(dst), src.numel(), float(scale), stream_ptr()), "m2f_quantize_fp8")
    return dst


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_pad: torch.Tensor, B: int, L: int, H: int,
                  drop_site: int = 0, drop_p: float = 0.0, rng: Optional[torch.Tensor] = None):
    """q/k/v: [B*L, H*hd] (possibly column slices).  Returns (out [B*L, H*hd], probs^T [B*H, Lp, Lp])."""
    runtime.require_gpu()
    E = q.shape[1]
    hd = E // H
    out = torch.empty(B * L, E, dtype=torch.float32, device=q.device)
    Lp = 16 * ((L + 15) // 16)
    probs = torch.zeros(B * H, Lp, Lp, dtype=torch.float32, device=q.device)
    kp = key_pad.to(torch.uint8).contiguous()
    check(lib().m2f_attention_fwd(B, L, H, hd, ptr(q), _ld(q), ptr(k), _ld(k), ptr(v), _ld(v), ptr(kp), ptr(out),
                                  _ld(out), ptr(probs), drop_site, drop_p, ptr(rng), stream_ptr()), "m2f_attention_fwd")
    return out, probs


def attention_bwd(q, k, v, key_pad, out, probs, dout, B: int, L: int, H: int, drop_site: int = 0, drop_p: float = 0.0,
                  rng: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    runtime.require_gpu()
    E = q.shape[1]
    hd = E // H
    dq, dk, dv = (torch.zeros(B * L, E, dtype=torch.float32, device=q.device) for _ in range(3))
    kp = key_pad.to(torch.uint8).contiguous()
    check(lib().m2f_attention_bwd(B, L, H, hd, ptr(q), _ld(q), ptr(k), _ld(k), ptr(v), _ld(v), ptr(kp), ptr(out),
                                  _ld(out), ptr(probs), ptr(dout), _ld(dout), ptr(dq), _ld(dq), ptr(dk), _ld(dk),
                                  ptr(dv), _ld(dv), drop_site, drop_p, ptr(rng), stream_ptr()), "m2f_attention_bwd")
    return dq, dk, dv


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, res: Optional[torch.Tensor] = None,
                  eps: float = 1e-5):
    runtime.require_gpu()
    T, d = x.shape
    out = torch.empty_like(x)
    stats = torch.empty(T, 2, dtype=torch.float32, device=x.device)
    check(lib().m2f_layernorm_fwd(T, d, ptr(x), ptr(gamma), ptr(beta), ptr(res), ptr(out), ptr(stats), eps,
                                  stream_ptr()), "m2f_layernorm_fwd")
    return out, stats


def layernorm_bwd(x, gamma, stats, dy, extra: Optional[torch.Tensor] = None):
    runtime.require_gpu()
    T, d = x.shape
    dx = torch.empty_like(x)
    partial = torch.empty((T + 3) // 4, 2, d, dtype=torch.float32, device=x.device)
    dg = torch.empty(d, dtype=torch.float32, device=x.device)
    db = torch.empty(d, dtype=torch.float32, device=x.device)
    check(lib().m2f_layernorm_bwd(T, d, ptr(x), ptr(gamma), ptr(stats), ptr(dy), ptr(extra), ptr(dx), ptr(partial),
                                  ptr(dg), ptr(db), stream_ptr()), "m2f_layernorm_bwd")
    return dx, dg, db


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, class_w: Optional[torch.Tensor] = None,
                  label_smoothing: float = 0.1, normalise: bool = True):
    """logits [T, C], labels int64 [T] (-1 = ignore) -> (loss_out[4] = loss, den, num, -; dlogits [T, C])."""
    runtime.require_gpu()
    T, C = logits.shape
    terms = torch.empty(T, 2, dtype=torch.float32, device=logits.device)
    dl = torch.empty(T, C, dtype=torch.float32, device=logits.device)
    out = torch.zeros(4, dtype=torch.float32, device=logits.device)
    check(lib().m2f_cross_entropy(T, C, ptr(logits.contiguous()), ptr(labels.contiguous()), ptr(class_w),
                                  label_smoothing, int(normalise), ptr(terms), ptr(dl), ptr(out), stream_ptr()),
          "m2f_cross_entropy")
    return out, dl


def fam_layer_forward(text, audio, key_pad, in_w, in_b, out_w, out_b, lin_w, lin_b, n_head: int,
                      precision: int = runtime.F32) -> torch.Tensor:
    """FusionAttentionModule.forward (reference src/model.py:13-20), dropout = identity."""
    B, L, E = text.shape
    t = text.reshape(B * L, E).contiguous()
    a = audio.reshape(B * L, E).contiguous()
    q = gemm(t, in_w[:E], NT, precision, bias=in_b[:E])
    k = gemm(a, in_w[E:2 * E], NT, precision, bias=in_b[E:2 * E])
    v = gemm(t, in_w[2 * E:], NT, precision, bias=in_b[2 * E:])
    att, _ = attention_fwd(q, k, v, key_pad.reshape(-1), B, L, n_head)
    x = gemm(att, out_w, NT, precision, bias=out_b)
    y = gemm(x, lin_w[:, :E], NT, precision, a1=t, b1=lin_w[:, E:], bias=lin_b, relu_a=True, relu_out=True)
    return y.view(B, L, E)
