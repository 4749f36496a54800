"""Deterministic synthetic weights / inputs for the parity fixtures.

Weights are NOT stored in the fixtures (a full-width state_dict is 50-400 MB); they are
regenerated bit-for-bit from (config, seed) with numpy's Philox bit generator, whose stream is
version-stable.  ``make_golden.py`` loads the same tensors into the real reference model via
``load_state_dict``; the tests regenerate them and feed the oracle and the HIP path.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import mer_amd.layout as layout  # noqa: E402

# name -> reference-style `config.model` dict.  Small widths carry full gradients in the fixture,
# full-width ones carry logits / loss / per-tensor gradient digests.
def _cfg(d_a, d_t, d_f, h_a, h_t, h_f, nl_a, nl_t, nl_f, nt_a=1, nt_t=1, hid=None, ncls=2,
         a_on=True, t_on=True, f_on=True, dropout=0.0):
    return {
        "dropout": dropout,
        "AUDIO": {"enabled": a_on, "embedding_size": d_a, "n_head": h_a, "n_transformers": nt_a, "n_encoder_layers": nl_a},
        "TEXT": {"enabled": t_on, "embedding_size": d_t, "n_head": h_t, "n_transformers": nt_t, "n_encoder_layers": nl_t},
        "FAM": {"enabled": f_on, "embedding_size": d_f, "n_head": h_f, "n_layers": nl_f},
        "CLASSIFIER": {"hidden_size": hid or d_f, "output_size": 7, "n_layers": ncls},
    }


CASES = {
    # name: (model config, B, L, lengths or None (= all full), input kind)
    "tiny_ragged": (_cfg(48, 64, 64, 4, 4, 4, 2, 2, 2), 5, 9, [9, 1, 4, 7, 2], "randn"),
    "tiny_shared_norm": (_cfg(64, 32, 48, 2, 4, 3, 1, 2, 1, nt_a=2, nt_t=3, hid=40, ncls=4), 3, 6, [6, 3, 5], "randn"),
    "tiny_odd_heads": (_cfg(60, 72, 96, 4, 3, 2, 1, 1, 3, hid=50, ncls=3), 4, 33, [33, 17, 1, 20], "randn"),
    "tiny_audio_only": (_cfg(64, 64, 64, 4, 4, 4, 2, 1, 1, t_on=False, f_on=False), 3, 5, [5, 2, 3], "randn"),
    "tiny_text_only": (_cfg(64, 64, 64, 4, 8, 4, 1, 2, 1, a_on=False, f_on=False), 2, 8, [8, 6], "randn"),
    "tiny_no_fam": (_cfg(32, 64, 64, 4, 4, 4, 1, 1, 1, f_on=False), 3, 7, [7, 7, 2], "randn"),
    # BASELINE.json configs[0] = C1: 1 enc layer + 1 FAM, 768/512/768, 4 dialogues x 16 utterances
    "c1": (_cfg(512, 768, 768, 8, 8, 8, 1, 1, 1), 4, 16, [16, 9, 12, 3], "randn"),
    # one full-width slice of the shipped config on REAL embedding rows (text_base + wav2vec2 val.pkl)
    "real_768_1layer": (_cfg(768, 768, 768, 8, 8, 8, 1, 1, 1), 6, 14, [14, 10, 3, 8, 1, 12], "real"),
    # audio_mel width (300) with a legal head count, shipped-depth slice kept shallow for size
    "c2_slice": (_cfg(300, 768, 768, 4, 8, 8, 2, 2, 2), 4, 16, [16, 16, 5, 11], "randn"),
    # BASELINE.json configs[2] = C3 geometry (roberta-large 1024 + wav2vec2 768, 8 heads -> head dims 128 / 96), depth 2,
    # at both dialogue lengths SURVEY 8-d names (16 and 24 utterances), ragged
    "c3_slice_l16": (_cfg(768, 1024, 768, 8, 8, 8, 2, 2, 2), 8, 16, [16, 16, 9, 3, 12, 16, 1, 7], "randn"),
    "c3_slice_l24": (_cfg(768, 1024, 768, 8, 8, 8, 2, 2, 2), 8, 24, [24, 17, 9, 24, 2, 13, 20, 5], "randn"),
}
FULL_GRAD_CASES = {k for k in CASES if k.startswith("tiny")}


def _gen(seed: int, idx: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, idx]))


def make_state_dict(model_cfg: dict, seed: int = 7) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state_dict in reference key order; aliased keys share one tensor."""
    c = layout.M2FConfig.from_model_config(model_cfg)
    specs, _ = layout.param_specs(c)
    out: Dict[str, torch.Tensor] = {}
    for i, sp in enumerate(specs):
        if sp.alias_of:
            out[sp.name] = out[sp.alias_of]
            continue
        g = _gen(seed, i)
        u = g.random(sp.shape, dtype=np.float32) * 2.0 - 1.0
        if sp.kind in ("linear_w", "attn_in_w"):
            w = u * np.float32(1.0 / math.sqrt(sp.fan_in))
        elif sp.kind == "ln_w":
            w = np.float32(1.0) + np.float32(0.1) * u
        elif sp.kind == "ln_b":
            w = np.float32(0.1) * u
        else:  # biases: non-zero so every bias path is exercised
            w = np.float32(0.05) * u
        out[sp.name] = torch.from_numpy(np.ascontiguousarray(w.astype(np.float32)))
    return out


def make_inputs(model_cfg: dict, B: int, L: int, lengths, kind: str = "randn", seed: int = 11,
                real_tables=None):
    """text [B,L,d_t], audio [B,L,d_a] (zero on pads), key_pad bool [B,L], emotion int64 [B,L] (-1 pads)."""
    d_t = model_cfg["TEXT"]["embedding_size"]
    d_a = model_cfg["AUDIO"]["embedding_size"]
    lengths = [L] * B if lengths is None else list(lengths)
    assert len(lengths) == B and max(lengths) == L and min(lengths) >= 1
    g = _gen(seed, 1000)
    if kind == "real":
        text_tab, audio_tab = real_tables
        rows = g.permutation(text_tab.shape[0])[: B * L].reshape(B, L)
        text = text_tab[torch.from_numpy(rows)].clone()
        audio = audio_tab[torch.from_numpy(rows)].clone()
    else:
        text = torch.from_numpy((g.standard_normal((B, L, d_t), dtype=np.float32) * np.float32(0.63)))
        audio = torch.from_numpy((g.standard_normal((B, L, d_a), dtype=np.float32) * np.float32(0.23)))
    emotion = torch.from_numpy(_gen(seed, 1001).integers(0, 7, size=(B, L)).astype(np.int64))  # own stream: kind-independent
    for b, n in enumerate(lengths):
        text[b, n:] = 0
        audio[b, n:] = 0
        emotion[b, n:] = -1
    key_pad = emotion == -1
    return text.contiguous(), audio.contiguous(), key_pad, emotion


def digest_vector(shape, seed: int, idx: int) -> torch.Tensor:
    """Seeded +-1 probe used for per-tensor gradient digests <grad, probe>."""
    g = _gen(seed ^ 0x5EED, idx)
    return torch.from_numpy((g.integers(0, 2, size=shape).astype(np.float32) * 2 - 1))


CLASS_WEIGHTS = torch.tensor([0.30, 1.20, 2.10, 1.30, 1.20, 5.30, 5.20], dtype=torch.float32)


# The reference's shipped model section (/root/reference/src/config.yaml:31-54, verbatim) = C2' of SURVEY 8
C2P_MODEL = {
    "dropout": 0.4,
    "AUDIO": {"enabled": True, "embedding_size": 768, "n_head": 8, "n_transformers": 1, "n_encoder_layers": 6},
    "TEXT": {"enabled": True, "embedding_size": 768, "n_head": 8, "n_transformers": 1, "n_encoder_layers": 6},
    "FAM": {"enabled": True, "embedding_size": 768, "n_head": 8, "n_layers": 5},
    "CLASSIFIER": {"hidden_size": 768, "output_size": 7, "n_layers": 2},
}


def c2p_real_val_case(fx):
    """The shipped-depth case on all 1,108 real val.pkl rows (make_golden.py::c2p_real_val_fixture):
    -> (model cfg, state_dict, batches) with batches = [(text [B,L,768], audio [B,L,768], key_pad [B,L], emotion [B,L], row_index [B,L])]
    in the validation loader's order (32 dialogues per batch, last one partial); row_index maps a slot to its row of
    fx['logits'] (-1 on pads).  Weights: make_state_dict, with the fitted last Linear the fixture carries."""
    cfg = C2P_MODEL
    sd = make_state_dict(cfg)
    names = list(sd.keys())
    sd[names[-2]] = torch.from_numpy(np.ascontiguousarray(fx["last_weight"]))
    sd[names[-1]] = torch.from_numpy(np.ascontiguousarray(fx["last_bias"]))
    text_tab, audio_tab = torch.from_numpy(fx["text_rows"]), torch.from_numpy(fx["audio_rows"])
    labels = torch.from_numpy(fx["labels"])
    lens = [int(x) for x in fx["lengths"]]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    batches = []
    for b0 in range(0, len(lens), 32):
        ls, ss = lens[b0: b0 + 32], starts[b0: b0 + 32]
        B, L = len(ls), max(ls)
        text, audio = torch.zeros(B, L, text_tab.shape[1]), torch.zeros(B, L, audio_tab.shape[1])
        emo = torch.full((B, L), -1, dtype=torch.int64)
        rows = torch.full((B, L), -1, dtype=torch.int64)
        for i, (n, s0) in enumerate(zip(ls, ss)):
            s0 = int(s0)
            text[i, :n], audio[i, :n] = text_tab[s0: s0 + n], audio_tab[s0: s0 + n]
            emo[i, :n] = labels[s0: s0 + n]
            rows[i, :n] = torch.arange(s0, s0 + n)
        batches.append((text, audio, emo == -1, emo, rows))
    return cfg, sd, batches
