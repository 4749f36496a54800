"""Model configuration and flat parameter layout of the M2FNet hot path.

The layout reproduces the reference's ``state_dict`` key order (SURVEY.md §8-b;
/root/reference/src/model.py:24-100) so checkpoints interchange with the reference.
All parameters live in ONE flat fp32 buffer in HBM (unique tensors only; the final
LayerNorm shared by the ``n_transformers`` encoders of a modality is stored once and
aliased under every ``*_encoders.{e}.norm.*`` key, as in the reference where
``nn.TransformerEncoder`` keeps the un-cloned ``norm`` object, model.py:62-65).
The C side (csrc/plan.hip ``m2f_param_layout``) computes the same offsets; the two
are cross-checked at load time.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

DIM_FF = 2048      # torch's default dim_feedforward, inherited by the reference (model.py:61,73)
LN_EPS = 1e-5      # torch's default layer_norm_eps


def _get(cfg, name):
    if isinstance(cfg, dict):
        return cfg[name]
    return getattr(cfg, name)


@dataclass(frozen=True)
class M2FConfig:
    """Plain-value view of the reference's ``config.model`` sub-tree (src/config.yaml:31-54)."""
    audio_enabled: bool
    text_enabled: bool
    fam_enabled: bool
    d_audio: int
    d_text: int
    d_fam: int
    nhead_audio: int
    nhead_text: int
    nhead_fam: int
    nlayers_audio: int
    nlayers_text: int
    nlayers_fam: int
    ntrans_audio: int
    ntrans_text: int
    cls_hidden: int
    cls_out: int
    cls_layers: int
    dropout: float
    dim_ff: int = DIM_FF
    ln_eps: float = LN_EPS

    @staticmethod
    def from_model_config(cfg) -> "M2FConfig":
        A, T, F, C = _get(cfg, "AUDIO"), _get(cfg, "TEXT"), _get(cfg, "FAM"), _get(cfg, "CLASSIFIER")
        c = M2FConfig(
            audio_enabled=bool(_get(A, "enabled")), text_enabled=bool(_get(T, "enabled")),
            fam_enabled=bool(_get(F, "enabled")),
            d_audio=int(_get(A, "embedding_size")), d_text=int(_get(T, "embedding_size")),
            d_fam=int(_get(F, "embedding_size")),
            nhead_audio=int(_get(A, "n_head")), nhead_text=int(_get(T, "n_head")),
            nhead_fam=int(_get(F, "n_head")),
            nlayers_audio=int(_get(A, "n_encoder_layers")), nlayers_text=int(_get(T, "n_encoder_layers")),
            nlayers_fam=int(_get(F, "n_layers")),
            ntrans_audio=int(_get(A, "n_transformers")), ntrans_text=int(_get(T, "n_transformers")),
            cls_hidden=int(_get(C, "hidden_size")), cls_out=int(_get(C, "output_size")),
            cls_layers=int(_get(C, "n_layers")), dropout=float(_get(cfg, "dropout")))
        c.validate()
        return c

    def validate(self) -> None:
        # same two checks, same messages as the reference (src/model.py:32-35)
        if not self.audio_enabled and not self.text_enabled:
            raise ValueError("At least one of audio and text must be enabled!")
        if self.fam_enabled and not (self.audio_enabled and self.text_enabled):
            raise ValueError("Fusion Attention Module can only be used with both audio and text enabled!")
        # nn.MultiheadAttention's own check (embed_dim must be divisible by num_heads)
        for on, d, h, nm in ((self.audio_enabled, self.d_audio, self.nhead_audio, "AUDIO"),
                             (self.text_enabled, self.d_text, self.nhead_text, "TEXT"),
                             (self.fam_enabled, self.d_fam, self.nhead_fam, "FAM")):
            if on and (h <= 0 or d % h != 0):
                raise AssertionError(f"embed_dim must be divisible by num_heads ({nm}: {d} % {h})")

    @property
    def cls_in(self) -> int:
        return 2 * self.d_fam if (self.audio_enabled and self.text_enabled) else self.d_fam

    def as_model_config(self) -> dict:
        """Inverse of from_model_config (dict tree with the reference's key names)."""
        return {
            "dropout": self.dropout,
            "AUDIO": {"enabled": self.audio_enabled, "embedding_size": self.d_audio, "n_head": self.nhead_audio,
                      "n_transformers": self.ntrans_audio, "n_encoder_layers": self.nlayers_audio},
            "TEXT": {"enabled": self.text_enabled, "embedding_size": self.d_text, "n_head": self.nhead_text,
                     "n_transformers": self.ntrans_text, "n_encoder_layers": self.nlayers_text},
            "FAM": {"enabled": self.fam_enabled, "embedding_size": self.d_fam, "n_head": self.nhead_fam,
                    "n_layers": self.nlayers_fam},
            "CLASSIFIER": {"hidden_size": self.cls_hidden, "output_size": self.cls_out, "n_layers": self.cls_layers},
        }


@dataclass
class ParamSpec:
    name: str                 # state_dict key
    shape: Tuple[int, ...]
    offset: int               # element offset into the flat fp32 buffer
    kind: str                 # "linear_w" | "linear_b" | "attn_in_w" | "attn_in_b" | "attn_out_b" | "ln_w" | "ln_b"
    fan_in: int = 0
    alias_of: str = ""        # non-empty: duplicate key aliasing an earlier tensor (shared final norm)

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def _align(n: int, a: int = 64) -> int:
    """Every tensor starts on a 256-byte boundary (64 floats) so 16-byte vector loads are legal."""
    return (n + a - 1) // a * a


def param_specs(c: M2FConfig) -> Tuple[List[ParamSpec], int]:
    """(specs in reference state_dict order, total flat length in elements)."""
    specs: List[ParamSpec] = []
    off = 0

    def add(name, shape, kind, fan_in=0):
        nonlocal off
        sp = ParamSpec(name, tuple(shape), off, kind, fan_in)
        specs.append(sp)
        off = _align(off + sp.numel)
        return sp

    def encoders(prefix, d, n_trans, n_layers):
        shared = {}
        for e in range(n_trans):
            for l in range(n_layers):
                p = f"{prefix}.{e}.layers.{l}."
                add(p + "self_attn.in_proj_weight", (3 * d, d), "attn_in_w", d)
                add(p + "self_attn.in_proj_bias", (3 * d,), "attn_in_b")
                add(p + "self_attn.out_proj.weight", (d, d), "linear_w", d)
                add(p + "self_attn.out_proj.bias", (d,), "attn_out_b")
                add(p + "linear1.weight", (c.dim_ff, d), "linear_w", d)
                add(p + "linear1.bias", (c.dim_ff,), "linear_b", d)
                add(p + "linear2.weight", (d, c.dim_ff), "linear_w", c.dim_ff)
                add(p + "linear2.bias", (d,), "linear_b", c.dim_ff)
                add(p + "norm1.weight", (d,), "ln_w")
                add(p + "norm1.bias", (d,), "ln_b")
                add(p + "norm2.weight", (d,), "ln_w")
                add(p + "norm2.bias", (d,), "ln_b")
            for nm, kind in (("weight", "ln_w"), ("bias", "ln_b")):
                key = f"{prefix}.{e}.norm.{nm}"
                if nm in shared:
                    src = shared[nm]
                    specs.append(ParamSpec(key, src.shape, src.offset, kind, 0, alias_of=src.name))
                else:
                    shared[nm] = add(key, (d,), kind)

    if c.audio_enabled:
        encoders("audio_encoders", c.d_audio, c.ntrans_audio, c.nlayers_audio)
        add("audio_proj.weight", (c.d_fam, c.d_audio), "linear_w", c.d_audio)
        add("audio_proj.bias", (c.d_fam,), "linear_b", c.d_audio)
    if c.text_enabled:
        encoders("text_encoders", c.d_text, c.ntrans_text, c.nlayers_text)
        add("text_proj.weight", (c.d_fam, c.d_text), "linear_w", c.d_text)
        add("text_proj.bias", (c.d_fam,), "linear_b", c.d_text)
    if c.fam_enabled:
        E = c.d_fam
        for i in range(c.nlayers_fam):
            p = f"fusion_layers.{i}."
            add(p + "multihead_attention.in_proj_weight", (3 * E, E), "attn_in_w", E)
            add(p + "multihead_attention.in_proj_bias", (3 * E,), "attn_in_b")
            add(p + "multihead_attention.out_proj.weight", (E, E), "linear_w", E)
            add(p + "multihead_attention.out_proj.bias", (E,), "attn_out_b")
            add(p + "linear.weight", (E, 2 * E), "linear_w", 2 * E)
            add(p + "linear.bias", (E,), "linear_b", 2 * E)
    h = c.cls_hidden
    add("output_layer.0.weight", (h, c.cls_in), "linear_w", c.cls_in)
    add("output_layer.0.bias", (h,), "linear_b", c.cls_in)
    idx = 0
    for _ in range(max(c.cls_layers - 2, 0)):
        idx += 2
        add(f"output_layer.{idx}.weight", (h, h), "linear_w", h)
        add(f"output_layer.{idx}.bias", (h,), "linear_b", h)
    idx += 3
    add(f"output_layer.{idx}.weight", (c.cls_out, h), "linear_w", h)
    add(f"output_layer.{idx}.bias", (c.cls_out,), "linear_b", h)
    return specs, off


def param_count(c: M2FConfig) -> int:
    """Number of scalar parameters (unique tensors; equals the reference's sum(p.numel()))."""
    return sum(s.numel for s in param_specs(c)[0] if not s.alias_of)


def flops_per_slot(c: M2FConfig, L: int) -> Tuple[float, float]:
    """(forward, forward+backward) algorithmic FLOPs per token-slot, SURVEY.md §8 closed form."""
    def enc(d):
        return 8 * d * d + 4 * d * c.dim_ff + 4 * L * d
    fwd = 0.0
    if c.audio_enabled:
        fwd += c.ntrans_audio * c.nlayers_audio * enc(c.d_audio) + 2 * c.d_audio * c.d_fam
    if c.text_enabled:
        fwd += c.ntrans_text * c.nlayers_text * enc(c.d_text) + 2 * c.d_text * c.d_fam
    if c.fam_enabled:
        fwd += c.nlayers_fam * (12 * c.d_fam ** 2 + 4 * L * c.d_fam)
    fwd += 2 * c.cls_in * c.cls_hidden + max(c.cls_layers - 2, 0) * 2 * c.cls_hidden ** 2 + 2 * c.cls_hidden * c.cls_out
    first = (c.d_audio ** 2 if c.audio_enabled else 0) + (c.d_text ** 2 if c.text_enabled else 0)
    return fwd, 3 * fwd - 6 * first
