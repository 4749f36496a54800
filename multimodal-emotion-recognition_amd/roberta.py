"""In-loop text feature extractor on the MI355X kernels (BASELINE config C5).

The reference produces its text embeddings in a separate stage with ``transformers.RobertaModel(add_pooling_layer=False)``
(src/feature_extractors/text/model.py:16-21; ``last_hidden_state[:, 0, :]`` at text/embeddings.py:83).  ``RobertaEncoder``
runs that model's eval-mode forward on the same device buffers the M2FNet step uses, so token ids can be fed to the training
loop directly: embeddings + LayerNorm kernel, per layer one packed Q/K/V GEMM, the long-sequence attention kernel, and the
grouped-GEMM epilogues for bias / residual / exact GELU, LayerNorm kernels in between.  State-dict keys are transformers'
own, so ``load_state_dict(RobertaModel.state_dict())`` works.  Inference only (the reference never back-propagates through
the extractor inside the M2FNet loop); no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import functional as F
from . import runtime
from .runtime import check, lib, ptr, stream_ptr


def _get(cfg, name, default=None):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name, default)


# M2F_ROBERTA_FAT=1: the bf16 mode as rounds 1-3 ran it (every activation as fp32 + bf16, fp32-operand attention) - for A/B measurements
_FAT_BF16 = os.environ.get("M2F_ROBERTA_FAT", "0") == "1"


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class RobertaEncoder(torch.nn.Module):
    def __init__(self, config, precision: str = "bf16"):
        super().__init__()
        self.d = int(_get(config, "hidden_size"))
        self.n_layers = int(_get(config, "num_hidden_layers"))
        self.n_head = int(_get(config, "num_attention_heads"))
        self.inter = int(_get(config, "intermediate_size"))
        self.vocab = int(_get(config, "vocab_size"))
        self.max_pos = int(_get(config, "max_position_embeddings"))
        self.type_vocab = int(_get(config, "type_vocab_size", 1))
        self.pad_id = int(_get(config, "pad_token_id", 1))
        self.eps = float(_get(config, "layer_norm_eps", 1e-5))
        act = _get(config, "hidden_act", "gelu")
        if act != "gelu":
            raise ValueError(f"hidden_act={act!r}: only the exact 'gelu' of RoBERTa is implemented")
        if self.d % self.n_head or self.d % 8 or self.inter % 8:
            raise ValueError("hidden_size must be divisible by num_attention_heads; hidden / intermediate sizes by 8")
        self.hd = self.d // self.n_head
        if self.hd > 128 or self.d > 2048:
            raise ValueError("head dim <= 128 and hidden size <= 2048")
        assert precision in ("bf16", "fp32", "fp8")
        # fp8 (BASELINE C5): the four GEMMs of every layer on fp8 MFMA (OCP e4m3, fp32 accumulate); weights quantised once
        # with a per-tensor scale 448 / amax, activations with fixed scales (LayerNorm / attention outputs x16, GELU
        # outputs x8, saturating) - everything else (embeddings, attention, LayerNorm, residual stream) stays fp32
        self.fp8 = precision == "fp8"
        self.precision = runtime.F32 if precision == "fp32" else runtime.BF16
        d, Fi = self.d, self.inter
        P = torch.nn.Parameter
        z = torch.zeros
        self.embeddings = torch.nn.Module()
        self.embeddings.word_embeddings = torch.nn.Embedding(self.vocab, d, padding_idx=self.pad_id)
        self.embeddings.position_embeddings = torch.nn.Embedding(self.max_pos, d, padding_idx=self.pad_id)
        self.embeddings.token_type_embeddings = torch.nn.Embedding(self.type_vocab, d)
        self.embeddings.LayerNorm = torch.nn.LayerNorm(d, eps=self.eps)
        self.encoder = torch.nn.Module()
        self.encoder.layer = torch.nn.ModuleList()
        for _ in range(self.n_layers):
            lyr = torch.nn.Module()
            lyr.attention = torch.nn.Module()
            lyr.attention.self = torch.nn.Module()
            lyr.attention.self.query = torch.nn.Linear(d, d)
            lyr.attention.self.key = torch.nn.Linear(d, d)
            lyr.attention.self.value = torch.nn.Linear(d, d)
            lyr.attention.output = torch.nn.Module()
            lyr.attention.output.dense = torch.nn.Linear(d, d)
            lyr.attention.output.LayerNorm = torch.nn.LayerNorm(d, eps=self.eps)
            lyr.intermediate = torch.nn.Module()
            lyr.intermediate.dense = torch.nn.Linear(d, Fi)
            lyr.output = torch.nn.Module()
            lyr.output.dense = torch.nn.Linear(Fi, d)
            lyr.output.LayerNorm = torch.nn.LayerNorm(d, eps=self.eps)
            self.encoder.layer.append(lyr)
        del P, z
        self._packed = None          # per-layer packed weights + bf16 copies, rebuilt when parameters change
        self._packed_versions = None
        self._ws = {}                # (B, S) -> workspace

    # transformers' RobertaModel registers non-persistent / legacy buffers under these names; accept and ignore them
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids") and not k.endswith("token_type_ids")}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = None
        return out

    # ---- packed weights ---------------------------------------------------------------------------------------------
    def _versions(self):
        return tuple(p._version for p in self.parameters())

    def _pack(self):
        dev = self.embeddings.word_embeddings.weight.device
        if dev.type != "cuda":
            raise runtime.HipError("RobertaEncoder runs on an MI355X only (move it with .to('cuda')): no CPU fallback")
        bf16 = self.precision == runtime.BF16

        def sh(w):
            return w.detach().to(torch.bfloat16).contiguous() if bf16 else None
        layers = []
        for lyr in self.encoder.layer:
            a = lyr.attention
            wqkv = torch.cat([a.self.query.weight, a.self.key.weight, a.self.value.weight], 0).detach().contiguous()
            bqkv = torch.cat([a.self.query.bias, a.self.key.bias, a.self.value.bias], 0).detach().contiguous()
            ent = {"wqkv": wqkv, "bqkv": bqkv, "wo": a.output.dense.weight.detach().contiguous(), "bo": a.output.dense.bias.detach(),
                   "g1": a.output.LayerNorm.weight.detach(), "b1": a.output.LayerNorm.bias.detach(),
                   "wi": lyr.intermediate.dense.weight.detach().contiguous(), "bi": lyr.intermediate.dense.bias.detach(),
                   "wo2": lyr.output.dense.weight.detach().contiguous(), "bo2": lyr.output.dense.bias.detach(),
                   "g2": lyr.output.LayerNorm.weight.detach(), "b2": lyr.output.LayerNorm.bias.detach()}
            for k in ("wqkv", "wo", "wi", "wo2"):
                ent[k + "16"] = sh(ent[k]) if not self.fp8 else None
                if self.fp8:
                    scale = 448.0 / max(float(ent[k].abs().max()), 1e-12)
                    ent[k + "8"] = (ent[k] * scale).clamp_(-448.0, 448.0).to(torch.float8_e4m3fn).contiguous()
                    ent[k + "8s"] = scale
            layers.append(ent)
        self._packed = layers
        self._packed_versions = self._versions()

    def _workspace(self, B: int, S: int, dev):
        key = (B, S)
        w = self._ws.get(key)
        if w is None:
            T, d, Fi = B * S, self.d, self.inter
            # ("t" LAST: the bf16 mode leaves it outside the shadow map - its only reader is the LayerNorm kernel, which reads fp32;
            #  "qkv" FIRST: the fp8 mode maps nothing else - the packed projection is the one activation it keeps as bf16)
            sizes = {"qkv": T * 3 * d, "x": T * d, "ctx": T * d, "y1": T * d, "h": T * Fi, "t": T * d}
            total = sum((n + 63) // 64 * 64 for n in sizes.values())
            ws = torch.zeros(total, dtype=torch.float32, device=dev)
            ws16 = torch.zeros(total, dtype=torch.bfloat16, device=dev)
            views, views16, off = {}, {}, 0
            for name, n in sizes.items():
                cols = {"x": d, "qkv": 3 * d, "ctx": d, "t": d, "y1": d, "h": Fi}[name]
                views[name] = ws[off: off + n].view(T, cols)
                views16[name] = ws16[off: off + n].view(T, cols)
                off += (n + 63) // 64 * 64
            w = {"ws": ws, "ws16": ws16, "v": views, "v16": views16, "mapped": total - (T * d + 63) // 64 * 64, "mapped8": (T * 3 * d + 63) // 64 * 64, "stats": torch.empty(T, 2, dtype=torch.float32, device=dev)}
            if self.fp8:
                w["q8"] = {n: torch.empty(T, c, dtype=torch.float8_e4m3fn, device=dev) for n, c in (("x", d), ("ctx", d), ("y1", d), ("h", Fi))}
            self._ws[key] = w
        return w

    # ---- forward ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> last_hidden_state [B, S, d] (rows of padded tokens are don't-care, as in transformers)."""
        runtime.require_gpu()
        if self._packed is None or self._packed_versions != self._versions():
            self._pack()
        B, S = input_ids.shape
        if S + self.pad_id + 1 > self.max_pos:
            raise ValueError(f"sequence length {S} needs position ids up to {S + self.pad_id}, but max_position_embeddings = {self.max_pos}")
        dev = input_ids.device
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        T, d, H, hd = B * S, self.d, self.n_head, self.hd
        prec = self.precision
        w = self._workspace(B, S, dev)
        v, v16 = w["v"], w["v16"]
        bf16 = prec == runtime.BF16 and not self.fp8
        # bf16 mode (round 4): the packed projection, the attention context and the FFN hidden activation exist ONLY as bf16 (their one
        # reader is the next GEMM / the attention kernel, which stage bf16 anyway); the residual stream and the LayerNorm inputs stay fp32
        lean = bf16 and d % 8 == 0 and hd % 8 == 0 and not _FAT_BF16
        # fp8 mode (round 4): the packed projection only as bf16 (the attention kernel's operand), the attention context and the LayerNorm
        # results as e4m3 straight from the kernels that produce them (they are GEMM operands): no fp32 copies of those, no quantise passes
        lean8 = self.fp8 and d % 16 == 0 and hd % 16 == 0 and not _FAT_BF16
        if lean8:
            check(lib().m2f_set_shadow_map(ptr(w["ws"]), ptr(w["ws16"]), w["mapped8"]), "m2f_set_shadow_map")
        else:
            check(lib().m2f_set_shadow_map(ptr(w["ws"]) if bf16 else None, ptr(w["ws16"]) if bf16 else None,
                                           (w["mapped"] if lean else w["ws"].numel()) if bf16 else 0), "m2f_set_shadow_map")
        try:
            ids = input_ids.reshape(-1).to(torch.int64).contiguous()
            keep = input_ids.ne(self.pad_id).to(torch.int64)
            pos = (torch.cumsum(keep, dim=1) * keep + self.pad_id).reshape(-1).contiguous()     # create_position_ids_from_input_ids
            key_pad = attention_mask.eq(0).to(torch.uint8).contiguous()
            emb = self.embeddings
            check(lib().m2f_embed_layernorm(T, d, ptr(ids), ptr(pos), ptr(emb.word_embeddings.weight), ptr(emb.position_embeddings.weight),
                                            ptr(emb.token_type_embeddings.weight), ptr(emb.LayerNorm.weight), ptr(emb.LayerNorm.bias),
                                            self.eps, ptr(v["x"]), d, stream_ptr()), "m2f_embed_layernorm")

            ACT_SCALE = {"x": 16.0, "ctx": 16.0, "y1": 16.0, "h": 8.0}

            def linear(a_name, wkey, L, out, bias, res=None, act=0, out8=None, out16_only=False):
                if self.fp8:
                    sa = ACT_SCALE[a_name]
                    q8 = w["q8"][a_name]
                    if a_name != "h" and not (lean8 and a_name in fresh8):   # h arrives quantised from the FFN1 epilogue
                        F.quantize_fp8(v[a_name], sa, out=q8)
                    if lean8 and out16_only and not out8:
                        check(lib().m2f_set_shadow_only(1), "m2f_set_shadow_only")
                    try:
                        F.gemm_fp8(q8, L[wkey + "8"], 1.0 / (sa * L[wkey + "8s"]), bias=bias, res=res, activation=act, out=out,
                                   out8=w["q8"][out8] if out8 else None, out8_scale=ACT_SCALE[out8] if out8 else 1.0)
                    finally:
                        if lean8 and out16_only and not out8:
                            check(lib().m2f_set_shadow_only(0), "m2f_set_shadow_only")
                else:
                    if lean and out16_only:
                        check(lib().m2f_set_shadow_only(1), "m2f_set_shadow_only")
                    try:
                        F.gemm(v[a_name], L[wkey], F.NT, prec, bias=bias, res=res, relu_out=act, out=out,
                               shadows=(v16[a_name], None, L[wkey + "16"], None) if bf16 else None)
                    finally:
                        if lean and out16_only:
                            check(lib().m2f_set_shadow_only(0), "m2f_set_shadow_only")
            fresh8 = set()                   # fp8 mode: activations whose e4m3 copy the producing kernel wrote itself

            def layernorm(src, g, b, dst):
                if lean8:
                    check(lib().m2f_layernorm_fwd_out8(T, d, ptr(v[src]), ptr(g), ptr(b), None, ptr(v[dst]), ptr(w["stats"]), self.eps,
                                                       ptr(w["q8"][dst]), ACT_SCALE[dst], stream_ptr()), "m2f_layernorm_fwd_out8")
                    fresh8.add(dst)
                else:
                    check(lib().m2f_layernorm_fwd(T, d, ptr(v[src]), ptr(g), ptr(b), None, ptr(v[dst]), ptr(w["stats"]), self.eps,
                                                  stream_ptr()), "m2f_layernorm_fwd")
            for L in self._packed:
                linear("x", "wqkv", L, v["qkv"], L["bqkv"], out16_only=True)
                if lean8:
                    q16 = v16["qkv"]
                    check(lib().m2f_attention_long_fwd_bf16_out8(B, S, H, hd, ptr(q16), 3 * d, q16.data_ptr() + 2 * d, 3 * d,
                                                                 q16.data_ptr() + 4 * d, 3 * d, ptr(key_pad), None, None, ptr(w["q8"]["ctx"]),
                                                                 ACT_SCALE["ctx"], d, stream_ptr()), "m2f_attention_long_fwd_bf16_out8")
                    fresh8.add("ctx")
                elif lean:
                    q16 = v16["qkv"]
                    check(lib().m2f_attention_long_fwd_bf16(B, S, H, hd, ptr(q16), 3 * d, q16.data_ptr() + 2 * d, 3 * d,
                                                            q16.data_ptr() + 4 * d, 3 * d, ptr(key_pad), ptr(v16["ctx"]), None, d,
                                                            stream_ptr()), "m2f_attention_long_fwd_bf16")
                else:
                    qkv = v["qkv"]
                    check(lib().m2f_attention_long_fwd(B, S, H, hd, ptr(qkv), 3 * d, qkv.data_ptr() + 4 * d, 3 * d,
                                                       qkv.data_ptr() + 8 * d, 3 * d, ptr(key_pad), ptr(v["ctx"]), d, stream_ptr()),
                          "m2f_attention_long_fwd")
                linear("ctx", "wo", L, v["t"], L["bo"], res=v["x"])
                layernorm("t", L["g1"], L["b1"], "y1")
                linear("y1", "wi", L, v["h"], L["bi"], act=2, out8="h" if self.fp8 else None, out16_only=True)
                linear("h", "wo2", L, v["t"], L["bo2"], res=v["y1"])
                layernorm("t", L["g2"], L["b2"], "x")
            out = v["x"].view(B, S, d).clone()
        finally:
            check(lib().m2f_set_shadow_map(None, None, 0), "m2f_set_shadow_map")
        return out

    def cls_embeddings(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[CLS] row of every sequence = the utterance embedding the reference stores (text/embeddings.py:83)."""
        return self.forward(input_ids, attention_mask)[:, 0, :].contiguous()
