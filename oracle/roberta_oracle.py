"""CPU restatement of the text feature extractor's encoder (TEST INFRASTRUCTURE ONLY, like m2fnet_oracle.py).

The reference computes text embeddings with ``transformers.RobertaModel(add_pooling_layer=False)``
(src/feature_extractors/text/model.py:16-21) and keeps ``last_hidden_state[:, 0, :]`` ([CLS], text/embeddings.py:83).
The arithmetic lives in the third-party ``transformers`` package (5.15.0 here; the reference pins 4.x - same eval-mode
math): RobertaEmbeddings -> N x RobertaLayer (post-LN BERT block, exact-erf GELU) .  This file restates it with explicit
tensor ops so the HIP path can be checked op by op; it is pinned against the real ``RobertaModel`` (random weights - the
pretrained ones cannot be fetched offline) by tests/golden/make_golden_roberta.py -> tests/golden/roberta_*.npz.
State-dict keys are transformers' own (``embeddings.word_embeddings.weight``, ``encoder.layer.{i}.attention.self.query.weight`` ...).
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def position_ids(input_ids: torch.Tensor, pad_id: int) -> torch.Tensor:
    """transformers create_position_ids_from_input_ids: positions count the non-pad tokens, offset by pad_id."""
    mask = input_ids.ne(pad_id).to(torch.int64)
    return torch.cumsum(mask, dim=1) * mask + pad_id


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def embeddings(sd: Dict[str, torch.Tensor], input_ids: torch.Tensor, pad_id: int, eps: float) -> torch.Tensor:
    pos = position_ids(input_ids, pad_id)
    x = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][pos] \
        + sd["embeddings.token_type_embeddings.weight"][0]
    return layer_norm(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)


def layer(sd: Dict[str, torch.Tensor], i: int, x: torch.Tensor, attention_mask: torch.Tensor, n_head: int, eps: float) -> torch.Tensor:
    p = f"encoder.layer.{i}."
    B, S, d = x.shape
    hd = d // n_head

    def lin(name, t):
        return t @ sd[p + name + ".weight"].t() + sd[p + name + ".bias"]

    def heads(t):
        return t.view(B, S, n_head, hd).permute(0, 2, 1, 3)
    q, k, v = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(lin("attention.self.value", x))
    scores = q @ k.transpose(-1, -2) / math.sqrt(hd)
    scores = scores.masked_fill(attention_mask[:, None, None, :] == 0, float("-inf"))
    ctx = (torch.softmax(scores, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    y1 = layer_norm(lin("attention.output.dense", ctx) + x, sd[p + "attention.output.LayerNorm.weight"],
                    sd[p + "attention.output.LayerNorm.bias"], eps)
    h = gelu(lin("intermediate.dense", y1))
    return layer_norm(lin("output.dense", h) + y1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def forward(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """-> last_hidden_state [B, S, d] (rows of padded tokens are don't-care, as in transformers)."""
    x = embeddings(sd, input_ids, cfg["pad_token_id"], cfg["layer_norm_eps"])
    for i in range(cfg["num_hidden_layers"]):
        x = layer(sd, i, x, attention_mask, cfg["num_attention_heads"], cfg["layer_norm_eps"])
    return x


def cls_embeddings(sd, cfg, input_ids, attention_mask) -> torch.Tensor:
    return forward(sd, cfg, input_ids, attention_mask)[:, 0, :]
