"""Criterion and optimizer of the reference's train loop as HIP-backed drop-ins.

* ``M2FCrossEntropyLoss``  = ``torch.nn.CrossEntropyLoss(weight, ignore_index=-1, label_smoothing=0.1)`` as the
  reference builds it (src/train.py:41-52), computed by the fused CE kernel (value + gradient in one pass).
* ``FusedAdam``            = ``torch.optim.Adam(model.parameters(), lr, weight_decay)`` (src/train.py:56): coupled
  L2, bias-corrected; ONE kernel over the flat parameter / gradient / moment buffers instead of ~130 per-tensor
  updates.  ``state_dict()`` keeps torch.optim.Adam's format (per-parameter ``step`` / ``exp_avg`` /
  ``exp_avg_sq`` indexed in reference parameter order), so reference checkpoints load and vice versa.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import functional as F
from . import runtime


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d, target1d, weight, label_smoothing):
        out, dl = F.cross_entropy(logits2d, target1d, weight, label_smoothing, True)
        ctx.save_for_backward(dl)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        (dl,) = ctx.saved_tensors
        return dl * grad_out, None, None, None


class M2FCrossEntropyLoss(nn.Module):
    def __init__(self, weight: Optional[torch.Tensor] = None, ignore_index: int = -1, label_smoothing: float = 0.1):
        super().__init__()
        if ignore_index != -1:
            raise ValueError("the fused criterion implements ignore_index=-1 (reference src/train.py:48-50)")
        self.register_buffer("weight", weight)
        self.label_smoothing = float(label_smoothing)

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        # reference call: criterion(outputs.permute(0, 2, 1), emotion) with input [B, C, L], target [B, L]
        if input.dim() == 3:
            C = input.shape[1]
            logits = input.permute(0, 2, 1).reshape(-1, C)
        else:
            logits = input
        w = self.weight.to(device=logits.device, dtype=torch.float32) if self.weight is not None else None
        return _CEFunction.apply(logits.contiguous().float(), target.reshape(-1).contiguous(), w, self.label_smoothing)


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (coupled L2 weight decay, reference ``src/train.py:56``) as ONE kernel over the model's flat
    parameter / gradient / moment buffers.  Differences from torch worth knowing: every parameter of the model is updated every
    step - a parameter whose ``.grad`` is None is treated as having a zero gradient (it still receives weight decay and the
    moment decay; torch skips it), which never happens on the M2FNet path, where backward writes every gradient; there is one
    parameter group (one lr / betas / eps / weight_decay for the whole model)."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.model = model
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._engine = None
        self._m = self._v = None
        self._step = 0
        self.grad_scale: Optional[torch.Tensor] = None     # device scalar: g <- g / grad_scale (data parallel)
        self._hyper: Optional[torch.Tensor] = None         # fused steps: lr / bc1, betas, eps, weight decay, 1 / sqrt(bc2) on the device
        self.grads_bf16: Optional[torch.Tensor] = None     # bf16 [n_params]: step() reads THIS instead of the fp32 gradient buffer (a plan
                                                           # armed with runtime.Plan.grad_bf16 left its gradients there, rounded once)

    def _bind(self):
        eng = self.model.engine()
        if eng is not self._engine:
            old = {id(p): self.state.get(p) for p in self.param_groups[0]["params"]}
            self._engine = eng
            self._m = torch.zeros_like(eng.flat)
            self._v = torch.zeros_like(eng.flat)
            for (p, o, n, s) in eng.items:
                st = old.get(id(p))
                mv, vv = self._m[o: o + n].view(s), self._v[o: o + n].view(s)
                if st:
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    self.state[p] = {"step": st["step"], "exp_avg": mv, "exp_avg_sq": vv}
        return eng

    def _materialise_state(self, eng):
        for (p, o, n, s) in eng.items:
            if p not in self.state or not self.state[p]:
                self.state[p] = {"step": torch.tensor(float(self._step)),
                                 "exp_avg": self._m[o: o + n].view(s), "exp_avg_sq": self._v[o: o + n].view(s)}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._bind()
        g = self.param_groups[0]
        flat_grad = eng.ensure_grad()
        # fast path: the engine published its flat-buffer views as .grad (checked on the two end parameters);
        # otherwise gather foreign .grad tensors into the flat buffer first
        first, last = eng.items[0][0], eng.items[-1][0]
        if self.grads_bf16 is not None or eng.grad_bf16_buf is not None:      # the step left its gradients rounded to bf16 (M2FNet.set_grad_bf16)
            flat_grad = self.grads_bf16 if self.grads_bf16 is not None else eng.grad_bf16_buf
        elif not (first.grad is eng.grad_views[0] and last.grad is eng.grad_views[-1]):
            for (p, o, n, s), view in zip(eng.items, eng.grad_views):
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
        self._step += 1
        if eng.wshadow is not None:
            # bf16 mode: the update and the bf16 shadows (W, W^T) of every 2-D parameter in ONE pass - the forward then skips its
            # parameter casts (engine.shadows_fresh)
            runtime.adam_step_shadowed(eng.cfg, eng.flat, flat_grad, self._m, self._v, eng.wshadow, self._step, g["lr"], g["betas"],
                                       g["eps"], g["weight_decay"], self.grad_scale)
            eng.mark_shadows_fresh()
        else:
            runtime.adam_step(eng.flat, flat_grad, self._m, self._v, self._step, g["lr"], g["betas"], g["eps"],
                              g["weight_decay"], self.grad_scale)
        return loss

    @torch.no_grad()
    def prepare_fused(self, plan) -> bool:
        """Arms `plan` so that its NEXT ``step()`` is also THIS optimizer's step: the weight-gradient launch applies the update to
        the elements whose gradient it holds in registers, one more launch inside the same graph updates the rest (bf16 mode, one
        process; csrc/gemm_p8.h EPI 3).  Same arithmetic on the same gradients as ``step()`` - bit-identical parameters, moments and
        parameter shadows (tests/test_fused_adam_gpu.py) - but the weight gradients of the table's matrices never reach memory:
        their ``.grad`` keeps whatever it held.  Returns False (and changes nothing) when the plan cannot; call ``finish_fused``
        after the step."""
        eng = self._bind()
        if eng.wshadow is None or not plan.train or not getattr(plan, "shared_shadow", False):
            return False
        if getattr(plan, "_fused_bad", False):
            return False
        if self._hyper is None:
            self._hyper = torch.zeros(8, dtype=torch.float32, device=eng.flat.device)
        key = (self._m.data_ptr(), self._v.data_ptr(), eng.flat.data_ptr(), eng.wshadow.data_ptr(), self._hyper.data_ptr(),
               self.grad_scale.data_ptr() if self.grad_scale is not None else 0)
        if getattr(plan, "_fused_key", None) != key:
            try:
                eng.ensure_grad()
                plan.fused_adam_setup(eng.flat, self._m, self._v, eng.wshadow, self._hyper, self.grad_scale)
            except runtime.HipError as e:
                plan._fused_bad = True                     # (another table form, ...): the caller takes the two-launch path
                plan._fused_err = str(e)
                return False
            plan._fused_key = key
        g = self.param_groups[0]
        self._step += 1
        runtime.adam_hyper(self._hyper, self._step, g["lr"], g["betas"], g["eps"], g["weight_decay"])
        plan.fused_adam(True)
        return True

    def finish_fused(self, plan) -> None:
        """After the armed step: the kernels wrote every parameter and both bf16 shadows of every matrix."""
        plan.fused_adam(False)
        self._engine.mark_shadows_fresh()

    @torch.no_grad()
    def step_ranges(self, ranges, before_each=None, grads=None):
        """One optimizer step issued as several kernel launches over contiguous element ranges [(lo, hi), ...] of the
        flat buffers (hi clipped to the parameter count; lo, hi multiples of 4).  `before_each(i)` runs before range i
        is launched - the data-parallel path waits there for that range's all-reduce, so the update of one bucket
        overlaps the exchange of the next.  `grads`: gradient buffer to read instead of the engine's (same indexing;
        fp32 or bf16 - the reduced buffer of the bf16 exchange).  Ranges made of whole parameter tensors keep the bf16
        parameter shadows current (m2f_adam_step_shadowed_range); other ranges leave them to the next forward's casts."""
        eng = self._bind()
        g = self.param_groups[0]
        flat_grad = eng.ensure_grad() if grads is None else grads
        n = eng.flat.numel()
        self._step += 1
        ranges = [(lo, min(hi, n)) for (lo, hi) in ranges]
        # bf16 mode with the model-wide parameter shadows: ranges that start and end at parameter tensors (dp.GradReducer aligns its
        # buckets that way) go through the shadow-writing kernel, so the next forward needs no parameter casts under data parallelism
        # either; anything else updates the parameters only and the next forward re-casts
        starts = self._tensor_starts(eng)
        shadowed = eng.wshadow is not None and all(lo in starts and (hi >= n or hi in starts) for lo, hi in ranges if hi > lo)
        if not shadowed:
            eng.invalidate_shadows()
        for i, (lo, hi) in enumerate(ranges):
            if before_each is not None:
                before_each(i)
            if hi <= lo:
                continue
            if shadowed:
                runtime.adam_step_shadowed(eng.cfg, eng.flat, flat_grad, self._m, self._v, eng.wshadow, self._step, g["lr"], g["betas"],
                                           g["eps"], g["weight_decay"], self.grad_scale, first=lo, end=(-1 if hi >= n else hi))
            else:
                runtime.adam_step(eng.flat[lo:hi], flat_grad[lo:hi], self._m[lo:hi], self._v[lo:hi], self._step, g["lr"],
                                  g["betas"], g["eps"], g["weight_decay"], self.grad_scale)
        if shadowed:
            covered = sorted((lo, hi) for lo, hi in ranges if hi > lo)
            whole = bool(covered) and covered[0][0] == 0 and covered[-1][1] >= n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            if whole:
                eng.mark_shadows_fresh()
            else:
                eng.invalidate_shadows()

    def _tensor_starts(self, eng):
        if getattr(self, "_starts_of", None) is not eng:
            self._starts = frozenset(int(o) for (_, o, _, _) in eng.items)
            self._starts_of = eng
        return self._starts

    def state_dict(self):
        if self._engine is not None and self._step > 0:
            self._materialise_state(self._engine)
            for p in self.param_groups[0]["params"]:
                self.state[p]["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        eng = self.model.engine()
        self._engine = eng
        self._m = torch.zeros_like(eng.flat)
        self._v = torch.zeros_like(eng.flat)
        steps = [int(st["step"]) for st in self.state.values() if st and "step" in st]
        self._step = max(steps) if steps else 0
        for (p, o, n, s) in eng.items:
            st = self.state.get(p)
            if st:
                mv, vv = self._m[o: o + n].view(s), self._v[o: o + n].view(s)
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
