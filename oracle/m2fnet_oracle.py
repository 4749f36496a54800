"""CPU oracle for the M2FNet dialogue-level training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / CPU baseline.  The
product path (``multimodal-emotion-recognition_amd``) never falls back to it.

What it is: an explicit, op-by-op restatement (plain ``torch`` CPU fp32 tensor
arithmetic: matmul, exp, sum, sqrt) of the arithmetic the reference obtains from
``torch.nn`` modules.  No ``nn.Transformer*``, ``nn.MultiheadAttention``,
``nn.LayerNorm``, ``nn.Linear``, ``nn.CrossEntropyLoss`` or ``torch.optim`` object is
used here; the backward pass is ``torch.autograd`` applied to these explicit ops.

Parity pin: the reference ships no tests or golden vectors, so this
oracle is pinned by running the REAL reference (``/root/reference/src/model.py``
imported on CPU in the build container) on seeded inputs:
``tests/golden/make_golden.py`` writes the fixtures, ``tests/test_oracle.py`` checks
this file against them (and against the live reference when it is present).

Reference sites restated (paths relative to /root/reference):
  src/model.py:5-20     FusionAttentionModule           -> fam_layer
  src/model.py:102-145  M2FNet.forward                  -> forward
  src/model.py:61-65    nn.TransformerEncoder stack     -> encoder_stack / encoder_layer
  src/train.py:41-52    CrossEntropyLoss(ls=0.1, ignore=-1[, weight]) -> cross_entropy
  src/train.py:56       torch.optim.Adam (coupled L2)   -> adam_step
  src/train.py:260-272  per-batch accuracy / weighted-F1, mean over batches -> batch_metrics
  src/dataset.py:71-89  collate_fn padding contract     -> collate
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
LN_EPS = 1e-5          # torch default layer_norm_eps, inherited at src/model.py:61-62
DIM_FF = 2048          # torch default dim_feedforward, inherited at src/model.py:61


def _get(cfg, name):
    """Attribute-or-key access so both Munch-like objects and dicts work."""
    if isinstance(cfg, dict):
        return cfg[name]
    return getattr(cfg, name)


# --------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """y = x W^T + b with W stored [out, in] (nn.Linear convention)."""
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float = LN_EPS) -> Tensor:
    """Biased-variance LayerNorm over the last dim (nn.LayerNorm semantics)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * g + b


def attention(q: Tensor, k: Tensor, v: Tensor, key_pad: Tensor, n_head: int,
              return_probs: bool = False):
    """Multi-head scaled-dot-product attention on [B, L, E] tensors.

    key_pad: bool [B, L], True = padded key (gets -inf before softmax), the
    key_padding_mask / src_key_padding_mask convention of src/model.py:14,107.
    """
    B, L, E = q.shape
    hd = E // n_head
    qh = q.reshape(B, L, n_head, hd).permute(0, 2, 1, 3)
    kh = k.reshape(B, L, n_head, hd).permute(0, 2, 1, 3)
    vh = v.reshape(B, L, n_head, hd).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    s = s - s.max(dim=-1, keepdim=True).values
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, L, E)
    return (o, p) if return_probs else o


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
def encoder_layer(x: Tensor, sd: Dict[str, Tensor], pre: str, key_pad: Tensor, n_head: int) -> Tensor:
    """Post-LN TransformerEncoderLayer (norm_first=False, ReLU), dropout = identity.

    x <- LN1(x + SA(x)); x <- LN2(x + W2 relu(W1 x + b1) + b2)   
    """
    E = x.shape[-1]
    qkv = linear(x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"])
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    a = attention(q, k, v, key_pad, n_head)
    a = linear(a, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"])
    x = layer_norm(x + a, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    h = torch.relu(linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"]))
    h = linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return layer_norm(x + h, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def encoder_stack(x: Tensor, sd: Dict[str, Tensor], pre: str, key_pad: Tensor, n_head: int,
                  n_layers: int) -> Tensor:
    """nn.TransformerEncoder(layer, num_layers, norm): layers then the final LayerNorm."""
    for l in range(n_layers):
        x = encoder_layer(x, sd, f"{pre}layers.{l}.", key_pad, n_head)
    return layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])


def fam_layer(text: Tensor, audio: Tensor, sd: Dict[str, Tensor], pre: str, key_pad: Tensor,
              n_head: int, inter: Optional[dict] = None) -> Tensor:
    """FusionAttentionModule.forward (src/model.py:13-20): Q = V = text, K = audio."""
    E = text.shape[-1]
    w = sd[pre + "multihead_attention.in_proj_weight"]
    b = sd[pre + "multihead_attention.in_proj_bias"]
    q = linear(text, w[:E], b[:E])
    k = linear(audio, w[E:2 * E], b[E:2 * E])
    v = linear(text, w[2 * E:], b[2 * E:])
    a, p = attention(q, k, v, key_pad, n_head, return_probs=True)
    x = linear(a, sd[pre + "multihead_attention.out_proj.weight"],
               sd[pre + "multihead_attention.out_proj.bias"])
    y = torch.relu(linear(torch.relu(torch.cat((x, text), dim=2)),
                          sd[pre + "linear.weight"], sd[pre + "linear.bias"]))
    if inter is not None:
        inter.update(q=q, k=k, v=v, p=p, attn=a, x=x, y=y)
    return y


def forward(sd: Dict[str, Tensor], cfg, text: Tensor, audio: Tensor, key_pad: Tensor,
            inter: Optional[dict] = None) -> Tensor:
    """M2FNet.forward (src/model.py:102-145) with every dropout as identity
    (eval mode, or train mode with model.dropout = 0.0).

    text [B,L,d_t], audio [B,L,d_a] fp32; key_pad bool [B,L] (True = pad) -> logits [B,L,C].
    """
    A, Tx, F, C = _get(cfg, "AUDIO"), _get(cfg, "TEXT"), _get(cfg, "FAM"), _get(cfg, "CLASSIFIER")
    a_on, t_on, f_on = bool(_get(A, "enabled")), bool(_get(Tx, "enabled")), bool(_get(F, "enabled"))
    if not a_on and not t_on:
        raise ValueError("At least one of audio and text must be enabled!")       # src/model.py:32-33
    if f_on and not (a_on and t_on):
        raise ValueError("Fusion Attention Module can only be used with both audio and text enabled!")

    if a_on:
        for e in range(_get(A, "n_transformers")):                                 # src/model.py:106-107
            audio = audio + encoder_stack(audio, sd, f"audio_encoders.{e}.", key_pad,
                                          _get(A, "n_head"), _get(A, "n_encoder_layers"))
        audio = linear(audio, sd["audio_proj.weight"], sd["audio_proj.bias"])      # :111-113
    if t_on:
        for e in range(_get(Tx, "n_transformers")):                                # :118-119
            text = text + encoder_stack(text, sd, f"text_encoders.{e}.", key_pad,
                                        _get(Tx, "n_head"), _get(Tx, "n_encoder_layers"))
        text = linear(text, sd["text_proj.weight"], sd["text_proj.bias"])          # :123-125
    if inter is not None:
        inter["audio_proj"] = audio if a_on else None
        inter["text_proj"] = text if t_on else None

    if f_on:
        for i in range(_get(F, "n_layers")):                                       # :129-131
            li = {} if inter is not None else None
            text = fam_layer(text, audio, sd, f"fusion_layers.{i}.", key_pad, _get(F, "n_head"), li)
            if inter is not None:
                inter[f"fam{i}"] = li
        x = torch.cat((audio, text), dim=2)                                        # :134  (audio, text)
    elif a_on and t_on:
        x = torch.cat((audio, text), dim=2)
    else:
        x = text if t_on else audio

    n_cls = _get(C, "n_layers")                                                    # :89-100
    x = linear(x, sd["output_layer.0.weight"], sd["output_layer.0.bias"])
    idx = 0
    for _ in range(max(n_cls - 2, 0)):
        idx += 2
        x = linear(torch.relu(x), sd[f"output_layer.{idx}.weight"], sd[f"output_layer.{idx}.bias"])
    idx += 3
    return linear(torch.relu(x), sd[f"output_layer.{idx}.weight"], sd[f"output_layer.{idx}.bias"])


# --------------------------------------------------------------------------------------
# criterion / optimizer / metrics
# --------------------------------------------------------------------------------------
def cross_entropy(logits: Tensor, target: Tensor, class_weight: Optional[Tensor] = None,
                  label_smoothing: float = 0.1, ignore_index: int = -1) -> Tensor:
    """CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1[, weight]) of src/train.py:48-50
    on logits [B,L,C] vs target int64 [B,L] (the reference permutes to [B,C,L], same thing).

    unweighted: mean_valid( (1-e)*(-logp_y) + e*mean_c(-logp_c) )
    weighted:   [ sum_valid (1-e)*w_y*(-logp_y) + e*sum_c w_c*(-logp_c)/C ] / sum_valid w_y
    (closed form verified against nn.CrossEntropyLoss).
    """
    C = logits.shape[-1]
    z = logits.reshape(-1, C)
    t = target.reshape(-1)
    valid = t != ignore_index
    zc = z - z.max(dim=-1, keepdim=True).values
    logp = zc - torch.log(torch.exp(zc).sum(dim=-1, keepdim=True))
    ts = torch.where(valid, t, torch.zeros_like(t))
    w = torch.ones(C, dtype=z.dtype) if class_weight is None else class_weight
    nll = -(logp.gather(1, ts[:, None])[:, 0]) * w[ts]
    smooth = -(logp * w[None, :]).sum(dim=-1) / C
    per = (1.0 - label_smoothing) * nll + label_smoothing * smooth
    vf = valid.to(z.dtype)
    return (per * vf).sum() / (w[ts] * vf).sum()


def adam_step(params: Sequence[Tensor], grads: Sequence[Tensor], exp_avg: List[Tensor],
              exp_avg_sq: List[Tensor], step: int, lr: float = 5e-5, weight_decay: float = 0.01,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam with COUPLED L2 (g += wd*p), bias-corrected (src/train.py:56).
    `step` is the 1-based step count AFTER this update.  In-place on params/state."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
            g = g + weight_decay * p
            m.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v.mul_(beta2).add_(g * g, alpha=1.0 - beta2)
            denom = v.sqrt() / math.sqrt(bc2) + eps
            p.add_(-(lr / bc1) * (m / denom))


def weighted_f1(y_true: Tensor, y_pred: Tensor) -> float:
    """sklearn f1_score(average='weighted') restated: per-class F1 over the labels present
    in y_true or y_pred, weighted by true support (0/0 -> 0)."""
    labels = torch.unique(torch.cat((y_true, y_pred)))
    total, acc = 0, 0.0
    for c in labels.tolist():
        tp = int(((y_true == c) & (y_pred == c)).sum())
        fp = int(((y_true != c) & (y_pred == c)).sum())
        fn = int(((y_true == c) & (y_pred != c)).sum())
        sup = tp + fn
        f1 = 0.0 if (2 * tp + fp + fn) == 0 else 2.0 * tp / (2 * tp + fp + fn)
        acc += f1 * sup
        total += sup
    return acc / total if total else 0.0


def batch_metrics(logits: Tensor, target: Tensor, ignore_index: int = -1) -> Tuple[float, float]:
    """One batch's (accuracy, weighted-F1) as src/train.py:261-267 computes them."""
    pred = logits.argmax(dim=2)
    m = target != ignore_index
    yt, yp = target[m].flatten(), pred[m].flatten()
    return float((yt == yp).to(torch.float64).mean()), weighted_f1(yt, yp)


def epoch_metrics(batches: Sequence[Tuple[Tensor, Tensor]]) -> Tuple[float, float]:
    """UNWEIGHTED mean over batches of the per-batch scores (src/train.py:272, src/test.py:73)."""
    accs, f1s = zip(*(batch_metrics(lg, tg) for lg, tg in batches))
    return sum(accs) / len(accs), sum(f1s) / len(f1s)


def collate(dialogues: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """collate_fn / apply_padding (src/dataset.py:71-89, src/utils.py:15-31): zero-pad features to
    the longest dialogue of the batch, pad labels with -1, padding_mask = (emotion == -1)."""
    L = max(d["text"].shape[0] for d in dialogues)
    B = len(dialogues)
    text = torch.zeros(B, L, dialogues[0]["text"].shape[1])
    audio = torch.zeros(B, L, dialogues[0]["audio"].shape[1])
    emo = torch.full((B, L), -1, dtype=torch.int64)
    for i, d in enumerate(dialogues):
        n = d["text"].shape[0]
        text[i, :n], audio[i, :n] = d["text"], d["audio"]
        emo[i, :n] = torch.as_tensor(d["emotion"], dtype=torch.int64).reshape(-1)
    return {"text": text, "audio": audio, "padding_mask": emo == -1, "emotion": emo}


def loss_and_grads(sd: Dict[str, Tensor], cfg, text, audio, key_pad, target,
                   class_weight: Optional[Tensor] = None):
    """forward + criterion + backward; returns (logits, loss, {name: grad}) for unique tensors."""
    leaves: Dict[int, Tensor] = {}
    sd2 = {}
    for k, v in sd.items():
        if id(v) not in leaves:
            leaves[id(v)] = v.detach().clone().requires_grad_(True)
        sd2[k] = leaves[id(v)]
    logits = forward(sd2, cfg, text, audio, key_pad)
    loss = cross_entropy(logits, target, class_weight)
    loss.backward()
    grads = {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in sd2.items()}
    return logits.detach(), loss.detach(), grads
