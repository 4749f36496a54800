"""``M2FNet`` / ``FusionAttentionModule``: host-side mirrors of the reference's ``src/model.py``.

Same constructor arguments, ``forward`` signature, ``state_dict`` keys and default initialisation as
/root/reference/src/model.py:5-145, but the modules here only HOLD parameters (as views into one flat
fp32 buffer in HBM); every forward/backward FLOP runs in the gfx950 kernels behind ``runtime.Plan``.
No ``nn.Transformer*`` / ``nn.MultiheadAttention`` / ``nn.Linear`` forward is ever called.
"""
from __future__ import annotations

import collections
import copy
import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import runtime
from .layout import M2FConfig, param_specs


# ------------------------------------------------------------------------------------------------------
# parameter holders (names chosen so the state_dict keys equal the reference's)
# ------------------------------------------------------------------------------------------------------
class _LinearParams(nn.Module):
    """weight [out, in], bias [out]; default init of nn.Linear (kaiming_uniform(a=sqrt(5)) + fan-in bias)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features) if in_features > 0 else 0.0
        nn.init.uniform_(self.bias, -bound, bound)


class _NormParams(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _MHAParams(nn.Module):
    """in_proj_weight [3E, E] (xavier-uniform), in_proj_bias = 0, out_proj.{weight, bias = 0}:
    the parameter set and init order of nn.MultiheadAttention (out_proj is created before the
    xavier init of in_proj_weight, so the RNG is consumed in the same order as the reference)."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = _LinearParams(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _EncoderLayerParams(nn.Module):
    def __init__(self, d: int, n_head: int, dim_ff: int):
        super().__init__()
        self.self_attn = _MHAParams(d, n_head)
        self.linear1 = _LinearParams(d, dim_ff)
        self.linear2 = _LinearParams(dim_ff, d)
        self.norm1 = _NormParams(d)
        self.norm2 = _NormParams(d)


class _EncoderStack(nn.Module):
    """nn.TransformerEncoder(encoder_layer, num_layers, norm): the layers are deep copies of ONE template
    (identical initial weights), the final norm object is shared, not cloned (model.py:61-65)."""

    def __init__(self, template: _EncoderLayerParams, d: int, n_head: int, dim_ff: int, num_layers: int,
                 norm: _NormParams):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(template) for _ in range(num_layers)])
        self.norm = norm


class FusionAttentionModule(nn.Module):
    """Mirror of reference src/model.py:5-20.  Inside ``M2FNet`` it is a parameter holder (the fusion stack
    runs in the plan); called on its own it executes the same gfx950 kernels layer-wise (inference only)."""

    def __init__(self, embedding_size: int, n_head: int, dropout: float):
        super().__init__()
        self.multihead_attention = _MHAParams(embedding_size, n_head)
        self.linear = _LinearParams(2 * embedding_size, embedding_size)
        self.relu = nn.ReLU()
        self.embedding_size, self.n_head, self.dropout_p = embedding_size, n_head, dropout

    def forward(self, text: torch.Tensor, audio: torch.Tensor, key_padding_mask: torch.Tensor) -> torch.Tensor:
        from . import functional as F
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("standalone FusionAttentionModule.forward is inference-only; train it inside M2FNet "
                               "(wrap the call in torch.no_grad())")
        return F.fam_layer_forward(text, audio, key_padding_mask, self.multihead_attention.in_proj_weight,
                                   self.multihead_attention.in_proj_bias, self.multihead_attention.out_proj.weight,
                                   self.multihead_attention.out_proj.bias, self.linear.weight, self.linear.bias,
                                   self.n_head)


class _Anchor(torch.autograd.Function):
    """Connects the plan's forward/backward to autograd through ONE dummy leaf: ``loss.backward()``
    (reference src/train.py:230) reaches ``backward`` below, which runs the HIP backward launch list and
    publishes the flat gradient buffer as the parameters' ``.grad`` views."""

    @staticmethod
    def forward(ctx, anchor, model, plan):
        eng = model._engine
        if plan.cfg.dropout > 0.0 and plan.train:
            runtime.check(runtime.lib().m2f_rng_advance(eng.rng.data_ptr(), runtime.stream_ptr()), "m2f_rng_advance")
        logits = plan.forward()
        ctx.model, ctx.plan, ctx.version = model, plan, plan.version
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        if not plan.handle:
            raise RuntimeError("M2FNet: the plan holding the activations of this forward was destroyed")
        if plan.version != ctx.version:
            raise RuntimeError("M2FNet: the activations of this forward were overwritten by a later forward of the "
                               "same plan shape (more graphs were kept alive than the plan cache may hold - raise "
                               "M2F_MAX_PLANS / M2F_MAX_PLAN_BYTES, or call backward() before further forwards)")
        plan.set_dlogits(dlogits)
        plan.backward()
        plan.release()                                    # (a second backward through the same graph re-runs on the same buffers)
        ctx.model._engine.publish_grads()
        return None, None, None


class _Engine:
    """Device state of one M2FNet: flat parameter / gradient buffers, dropout RNG state, plan cache."""

    def __init__(self, model: "M2FNet", device: torch.device):
        runtime.require_gpu()
        self.model, self.device, self.cfg = model, device, model.m2f_config
        total = runtime.verify_layout(self.cfg)
        specs, _ = param_specs(self.cfg)
        named = dict(model.named_parameters(remove_duplicate=False))
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad: Optional[torch.Tensor] = None
        self.flat_grad_ext: Optional[torch.Tensor] = None
        self.items = []                       # (param, offset, numel, shape)
        seen = set()
        with torch.no_grad():
            for sp in specs:
                p = named[sp.name]
                if sp.alias_of or id(p) in seen:
                    continue
                seen.add(id(p))
                view = self.flat[sp.offset: sp.offset + sp.numel].view(sp.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                self.items.append((p, sp.offset, sp.numel, sp.shape))
        from .dp import dropout_seed
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        lo, hi = dropout_seed(torch.initial_seed(), rank)     # replicas share weights (same seed) but not dropout masks
        to_i32 = lambda u: u - (1 << 32) if u >= (1 << 31) else u
        self.rng = torch.tensor([to_i32(lo), to_i32(hi), 0, 0], dtype=torch.int32, device=device)
        # plan cache: least-recently-used first; at most `max_plans` plans / `max_plan_bytes` of workspaces (+ captured graphs)
        # stay alive.  Keys are (shape bucket, mode, instance): a shape whose plan still holds the activations of a forward
        # that has not run its backward yet (`Plan.busy`) gets a second instance instead of overwriting them, so
        # `forward(A); forward(B); loss_A.backward()` works as it does in the reference when A and B share a bucket.
        # Shape buckets make the working set of a training run 3 L-buckets x {full, last partial batch} x {train, eval}.
        self.plans: "collections.OrderedDict[Tuple, runtime.Plan]" = collections.OrderedDict()
        self.max_plans = int(os.environ.get("M2F_MAX_PLANS", "16"))
        self.max_plan_bytes = int(float(os.environ.get("M2F_MAX_PLAN_BYTES", str(96 * 2 ** 30))))
        self.shape_buckets = model.shape_buckets
        self.grad_views = None
        self.anchor = torch.zeros(1, device=device, requires_grad=True)
        self.stream = torch.cuda.Stream(device=device)    # hipGraph capture is illegal on the default stream
        self.precision = runtime.PRECISIONS[model.precision]
        # bf16 mode: ONE buffer of bf16 parameter shadows (W and W^T of every 2-D parameter) for all plans, kept current by
        # FusedAdam.step itself (m2f_adam_step_shadowed).  `_fresh_token` = the parameters' version counters at the moment the
        # optimizer last wrote the shadows: any later in-place change through torch (load_state_dict, a foreign optimizer,
        # p.mul_(), writes through the flat buffer or its views) moves the counters, the plans then re-cast the shadows at the head of
        # their forward as before.  Writes that bypass the counters (p.data...) need `invalidate_shadows()`; `flat_parameters()` calls it.  M2F_SHARED_SHADOWS=0: off.
        self.wshadow: Optional[torch.Tensor] = None
        self.grad_bf16_buf: Optional[torch.Tensor] = None     # bf16 [n_params]: train plans leave their gradients here (set_grad_bf16)
        self._fresh_token = None
        if self.precision == runtime.BF16 and os.environ.get("M2F_SHARED_SHADOWS", "1") != "0":
            self.wshadow = runtime.param_shadow_buffer(self.cfg, device)

    def _version_token(self):
        # the parameters' own counters + the flat buffer's (shared by every view of it: a write through `flat_parameters()` or a
        # slice of it moves that one; the fused Adam kernels write through raw pointers and move neither)
        # (an engine first built under torch.inference_mode() holds an inference tensor: no counter, and no in-place writes outside
        # inference mode either)
        return (sum(p._version for (p, _, _, _) in self.items), 0 if self.flat.is_inference() else self.flat._version)

    def mark_shadows_fresh(self) -> None:
        self._fresh_token = self._version_token()

    def invalidate_shadows(self) -> None:
        self._fresh_token = None

    def shadows_fresh(self) -> bool:
        return self.wshadow is not None and self._fresh_token is not None and self._fresh_token == self._version_token()

    def ensure_grad(self) -> torch.Tensor:
        if self.flat_grad is None:
            from .dp import TAIL
            # [gradients | den, num, 0...]: the tail rides along in the data-parallel all-reduce (dp.py)
            self.flat_grad_ext = torch.zeros(self.flat.numel() + TAIL, dtype=torch.float32, device=self.device)
            self.flat_grad = self.flat_grad_ext[: self.flat.numel()]
            self.grad_views = [self.flat_grad[o: o + n].view(s) for (_, o, n, s) in self.items]
        return self.flat_grad

    @staticmethod
    def bucket(B: int, L: int) -> Tuple[int, int]:
        """Plan shape for a batch of B dialogues x L utterances: L rounded up to a multiple of 16 (the attention kernels'
        tile; MELD batches have L anywhere in 1..33 -> three shapes), B to a power of two below 8 and a multiple of 8 above
        (only the last, partial batch of an epoch differs from batch_size)."""
        Lb = (L + 15) // 16 * 16
        Bb = 1 << max(B - 1, 0).bit_length() if B <= 8 else (B + 7) // 8 * 8
        return Bb, Lb

    def plan(self, B: int, L: int, want_backward: bool, dropout_active: bool, valid: Optional[int] = None) -> runtime.Plan:
        """valid: number of valid utterances of the batch (packed mode) - the plan then holds that many token rows (rounded up to
        a multiple of 64, plus one row per filler dialogue and one spare) instead of B x L slots; batches that are at least
        85 % full keep the padded plan."""
        b_in = B
        if self.shape_buckets:
            B, L = self.bucket(B, L)
        T = None
        if valid is not None:
            need = int(valid) + (B - b_in) + 1
            Tb = (need + 63) // 64 * 64
            if Tb <= 0.85 * B * L:
                T = max(Tb, B)
        base = (B, L, T, want_backward, dropout_active, self.precision)
        inst, key, pl, oldest = 0, None, None, None
        while True:                                       # first instance of this shape that no live autograd graph owns
            key = base + (inst,)
            pl = self.plans.get(key)
            if pl is None or not pl.busy():
                break
            oldest = oldest or key
            inst += 1
        if pl is None:
            if inst > 0 and not self._room_for(self.plans[oldest].nbytes()):
                # no room for another instance: hand out the least recently used one (its pending backward will raise)
                key = next(k for k in self.plans if k[:-1] == base)
                pl = self.plans[key]
        if pl is None:
            cfg = self.cfg
            if not dropout_active and cfg.dropout != 0.0:
                cfg = M2FConfig(**{**cfg.__dict__, "dropout": 0.0})
            # the C side couples "keeps a backward list" and "dropout active" in its train flag
            train = want_backward or dropout_active
            if train:
                self.ensure_grad()
            self._evict(max(self.max_plans, 1) - 1, self.max_plan_bytes)
            pl = runtime.Plan(cfg, B, L, self.precision, train, self.flat, self.flat_grad_ext if train else None, self.rng, T=T,
                              param_shadow=self.wshadow)
            pl._on_cast = self.mark_shadows_fresh
            self.plans[key] = pl
            self._evict(max(self.max_plans, 1), self.max_plan_bytes, protect=key)
        else:
            self.plans.move_to_end(key)
            self._evict(max(self.max_plans, 1), self.max_plan_bytes, protect=key)      # (the caps may have been lowered since)
        pl.params_fresh(self.shadows_fresh())
        if pl.train:
            self._arm_grad_bf16(pl)
        return pl

    # -- gradients left as bf16 by the step (M2FNet.set_grad_bf16) -------------------------------------------------------------
    def _arm_grad_bf16(self, pl) -> None:
        want = self.grad_bf16_buf
        if getattr(pl, "_g16_ref", None) is want or getattr(pl, "_g16_bad", False):
            return
        try:
            pl.grad_bf16(want)
        except runtime.HipError:
            if want is None:
                raise
            pl._g16_bad = True                               # (fp32 mode, another table form): this plan keeps fp32 gradients ...
            self.grad_bf16_buf = None                        # ... and then so does every plan: the optimizer reads ONE buffer
            for other in self.plans.values():
                if getattr(other, "_g16_ref", None) is not None:
                    other.grad_bf16(None)

    def _room_for(self, nbytes: int) -> bool:
        """Could one more plan of `nbytes` be cached after evicting every idle plan?"""
        busy = [p for p in self.plans.values() if p.busy()]
        return len(busy) + 1 <= max(self.max_plans, 1) and sum(p.nbytes() for p in busy) + nbytes <= self.max_plan_bytes

    def _evict(self, keep: int, keep_bytes: Optional[int] = None, protect=None) -> None:
        """Drop least-recently-used IDLE plans until at most `keep` plans / `keep_bytes` of workspaces are left: frees their
        workspaces and captured graphs.  A plan whose activations a live autograd graph still needs is never closed."""
        def over():
            return len(self.plans) > keep or (keep_bytes is not None and self.plan_bytes() > keep_bytes and len(self.plans) > 1)
        for k in list(self.plans):
            if not over():
                break
            if k == protect or self.plans[k].busy():
                continue
            old = self.plans.pop(k)
            torch.cuda.synchronize(self.device)                        # nothing queued may still use it
            old.close()

    def plan_bytes(self) -> int:
        """HBM held by the cached plans' workspaces."""
        return sum(p.nbytes() for p in self.plans.values())

    def publish_grads(self) -> None:
        """Expose the flat gradient buffer as ``p.grad`` views.  Gradients are OVERWRITTEN each backward
        (the reference zeroes them every step, src/train.py:227); a foreign ``.grad`` tensor is added to."""
        for (p, _, _, _), v in zip(self.items, self.grad_views):
            g = p.grad
            if g is None:
                p.grad = v
            elif g is not v and g.data_ptr() != v.data_ptr():
                g.add_(v)

    def owns(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for (p, o, _, _) in self.items)


class M2FNet(nn.Module):
    """Drop-in for reference ``src/model.py:23-145``: ``M2FNet(config.model)``; ``forward(text, audio, mask)``
    with text [B,L,d_t], audio [B,L,d_a] fp32 and mask bool [B,L] (True = pad) -> logits [B,L,output_size].

    Limits the reference does not have: at most 64 utterances per dialogue (L <= 64: the dialogue attention kernels keep a
    whole dialogue in one workgroup; MELD's longest dialogue has 33) - longer inputs raise from ``m2f_plan_create``; each
    backward OVERWRITES the gradients (the reference zeroes them every step, ``src/train.py:227``), so accumulating over
    several backward calls needs a caller-side buffer."""

    def __init__(self, config, precision: Optional[str] = None, shape_buckets: Optional[bool] = None,
                 packed: Optional[bool] = None):
        super().__init__()
        self.config = config
        # packed ("varlen") token layout for `train_step` and no-grad `forward` (runtime.Plan, m2f_plan_create_packed): a ragged
        # batch costs its valid utterances, not B x L slots; needs the batch's valid count on the host (one sync per call).
        # Off by default (M2F_PACKED=1 or packed=True): the reference's batches reach the model as padded tensors either way
        self.packed = (os.environ.get("M2F_PACKED", "0") == "1") if packed is None else bool(packed)
        # round batch shapes up to a few plan shapes (see _Engine.bucket); M2F_SHAPE_BUCKETS=0 plans every shape exactly
        self.shape_buckets = (os.environ.get("M2F_SHAPE_BUCKETS", "1") != "0") if shape_buckets is None else bool(shape_buckets)
        c = M2FConfig.from_model_config(config)           # raises the reference's two ValueErrors
        self.m2f_config = c
        self.audio_enabled, self.text_enabled, self.fam_enabled = c.audio_enabled, c.text_enabled, c.fam_enabled
        self.n_head_audio, self.n_head_text, self.n_head_fam = c.nhead_audio, c.nhead_text, c.nhead_fam
        self.dropout = nn.Dropout(c.dropout)               # kept for attribute parity; never called
        # GEMM operand precision: "fp32" (exact-fp32 MFMA, the 1e-3 parity mode) or "bf16" (bf16 MFMA, fp32 accumulate)
        self.precision = precision or os.environ.get("M2F_PRECISION", "fp32")
        if self.precision not in runtime.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(runtime.PRECISIONS)}")

        # construction order == reference (model.py:59-100) so that a given torch seed yields the same weights
        if c.audio_enabled:
            tmpl = _EncoderLayerParams(c.d_audio, c.nhead_audio, c.dim_ff)
            norm = _NormParams(c.d_audio)
            self.audio_encoders = nn.ModuleList([
                _EncoderStack(tmpl, c.d_audio, c.nhead_audio, c.dim_ff, c.nlayers_audio, norm)
                for _ in range(c.ntrans_audio)])
            self.audio_proj = _LinearParams(c.d_audio, c.d_fam)
        if c.text_enabled:
            tmpl = _EncoderLayerParams(c.d_text, c.nhead_text, c.dim_ff)
            norm = _NormParams(c.d_text)
            self.text_encoders = nn.ModuleList([
                _EncoderStack(tmpl, c.d_text, c.nhead_text, c.dim_ff, c.nlayers_text, norm)
                for _ in range(c.ntrans_text)])
            self.text_proj = _LinearParams(c.d_text, c.d_fam)
        if c.fam_enabled:
            self.fusion_layers = nn.ModuleList([
                FusionAttentionModule(embedding_size=c.d_fam, n_head=c.nhead_fam, dropout=c.dropout)
                for _ in range(c.nlayers_fam)])
        head = [_LinearParams(c.cls_in, c.cls_hidden)]
        for _ in range(max(c.cls_layers - 2, 0)):
            head.append(nn.ReLU())
            head.append(_LinearParams(c.cls_hidden, c.cls_hidden))
        head.append(nn.ReLU())
        head.append(self.dropout)
        head.append(_LinearParams(c.cls_hidden, c.cls_out))
        self.output_layer = nn.Sequential(*head)
        self._engine: Optional[_Engine] = None

    # -- device plumbing -------------------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        self._engine = None                                # .to()/.cuda()/.cpu() re-home the parameters
        return super()._apply(fn, *args, **kwargs)

    def engine(self, device: Optional[torch.device] = None) -> _Engine:
        if device is None:
            device = next(self.parameters()).device
        if device.type != "cuda":
            raise runtime.HipError("M2FNet runs only on an MI355X (gfx950): move the model and the batch to 'cuda' "
                                   "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests only)")
        if self._engine is None or self._engine.device != device or not self._engine.owns():
            self._engine = _Engine(self, device)
        return self._engine

    # -- reference surface -----------------------------------------------------------------------------
    def forward(self, text, audio, mask):
        eng = self.engine(mask.device)
        B, L = mask.shape
        want_bwd = torch.is_grad_enabled() and any(p.requires_grad for p, *_ in eng.items)
        valid = int((~mask.bool()).sum()) if self.packed else None
        plan = eng.plan(B, L, want_bwd, self.training and self.m2f_config.dropout > 0.0, valid)
        plan.set_inputs(text if self.text_enabled else None, audio if self.audio_enabled else None, mask)
        if want_bwd:
            out = _Anchor.apply(eng.anchor, self, plan)
            plan.hold(out.grad_fn)
            return out
        if plan.train and plan.cfg.dropout > 0.0:
            runtime.check(runtime.lib().m2f_rng_advance(eng.rng.data_ptr(), runtime.stream_ptr()), "m2f_rng_advance")
        return plan.forward().clone()

    # -- fused fast path (forward + criterion + backward as one launch list / hipGraph) ---------------
    def train_step(self, text, audio, mask, emotion, label_smoothing: float = 0.1,
                   class_weights: Optional[torch.Tensor] = None, normalise: bool = True,
                   use_graph: bool = True, optimizer=None) -> torch.Tensor:
        """Body of reference src/train.py:227-230 in one call: returns the (device) loss scalar and leaves
        the gradients in ``p.grad`` (views of the flat buffer).
        optimizer (a ``FusedAdam`` of this model): the call is ALSO ``optimizer.step()`` (src/train.py:231) - in bf16 mode the
        weight-gradient launch applies the update itself (``FusedAdam.prepare_fused``; the matrices' ``.grad`` is then not written),
        otherwise the optimizer's own kernel runs behind the step."""
        eng = self.engine(mask.device)
        B, L = mask.shape
        valid = int((~mask.bool()).sum()) if self.packed else None
        plan = eng.plan(B, L, True, self.training and self.m2f_config.dropout > 0.0, valid)

        def body():
            plan.set_inputs(text if self.text_enabled else None, audio if self.audio_enabled else None, mask, emotion)
            if class_weights is not None:
                plan.class_w[: class_weights.numel()].copy_(class_weights)
            if optimizer is None:
                return plan.step(label_smoothing, class_weights is not None, normalise, use_graph)
            plan.params_fresh(eng.shadows_fresh())
            if optimizer.prepare_fused(plan):
                out = plan.step(label_smoothing, class_weights is not None, normalise, use_graph)
                optimizer.finish_fused(plan)
                return out
            out = plan.step(label_smoothing, class_weights is not None, normalise, use_graph)
            eng.publish_grads()
            optimizer.step()
            return out

        if use_graph:                                    # capture / replay on the engine's own stream
            cur = torch.cuda.current_stream(eng.device)
            eng.stream.wait_stream(cur)
            with torch.cuda.stream(eng.stream):
                loss = body()
            cur.wait_stream(eng.stream)
        else:
            loss = body()
        eng.publish_grads()
        return loss[0].clone()          # (the buffer is overwritten by the next step)

    def set_grad_bf16(self, on: bool = True) -> bool:
        """bf16 mode: every following training step leaves its gradients ROUNDED ONCE TO BF16 in one flat bf16 buffer - the weight-gradient
        launch writes bf16 dW directly, one cast launch rounds the rest - and ``FusedAdam`` reads that buffer (fp32 moments and parameters as
        ever).  It is the precision every rank's gradient has under the data-parallel bf16 exchange (dp.py), so a one-GPU run and an
        eight-GPU run then train with the same gradient precision; it saves the fp32 dW round trip (8 bytes per parameter and step: 2.66 ->
        2.59 ms per C3 step).  The matrices' fp32 ``.grad`` is NOT written in this mode.  Returns whether the mode is on (fp32 models: no)."""
        eng = self.engine()
        if not on:
            eng.grad_bf16_buf = None
        elif eng.precision == runtime.BF16 and eng.grad_bf16_buf is None:
            eng.grad_bf16_buf = torch.zeros(eng.flat.numel(), dtype=torch.bfloat16, device=eng.flat.device)
        for pl in list(eng.plans.values()):
            if pl.train:
                eng._arm_grad_bf16(pl)
        return eng.grad_bf16_buf is not None

    def invalidate_shadows(self) -> None:
        """Call after writing parameters in a way torch's version counters do not see (``p.data`` edits, writes through
        ``flat_parameters()``): the next forward re-casts the bf16 parameter shadows."""
        if self._engine is not None:
            self._engine.invalidate_shadows()

    def flat_parameters(self) -> torch.Tensor:
        """The flat fp32 parameter buffer (reference state_dict order).  Handing it out invalidates the bf16 parameter shadows:
        the caller may write through it (in-place writes also move its version counter, which the freshness token includes;
        writes that bypass the counters - ``.data`` - are caught by this call having invalidated)."""
        eng = self.engine()
        eng.invalidate_shadows()
        return eng.flat

    def flat_gradients(self) -> torch.Tensor:
        return self.engine().ensure_grad()
