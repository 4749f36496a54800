"""ctypes binding of ``csrc/libm2fnet_hip.so`` (C ABI: ``include/m2fnet_hip.h``).

PyTorch is used for plumbing only: device memory (tensors), the current HIP stream and, in ``dp.py``,
``torch.distributed`` (RCCL).  All arithmetic of the hot path happens in the HIP kernels behind this
binding.  There is NO fallback: if the shared library is missing or the device is not gfx950 the import /
first use raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional, Tuple

import torch

from .layout import M2FConfig, param_specs

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("M2F_LIB", os.path.join(CSRC, "libm2fnet_hip.so"))   # M2F_LIB: experiment builds only
HEADER_PATH = os.path.abspath(os.path.join(_HERE, "..", "include", "m2fnet_hip.h"))

F32, BF16 = 0, 1
PRECISIONS = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16}
(BUF_TEXT, BUF_AUDIO, BUF_KEYPAD, BUF_LABELS, BUF_CLASSW, BUF_LOGITS, BUF_LOSS, BUF_DLOGITS,
 BUF_FAM0_OUT, BUF_CU_SEQLENS) = range(10)

c_void_p, c_int, c_float, c_int64, c_uint32 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_int64, ctypes.c_uint32)


class M2FConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "audio_enabled", "text_enabled", "fam_enabled", "d_audio", "d_text", "d_fam",
        "nhead_audio", "nhead_text", "nhead_fam", "nlayers_audio", "nlayers_text", "nlayers_fam",
        "ntrans_audio", "ntrans_text", "cls_hidden", "cls_out", "cls_layers", "dim_ff")] + [
        ("dropout", ctypes.c_float), ("ln_eps", ctypes.c_float)]


def config_to_c(c: M2FConfig) -> M2FConfigC:
    return M2FConfigC(int(c.audio_enabled), int(c.text_enabled), int(c.fam_enabled), c.d_audio, c.d_text, c.d_fam,
                      c.nhead_audio, c.nhead_text, c.nhead_fam, c.nlayers_audio, c.nlayers_text, c.nlayers_fam,
                      c.ntrans_audio, c.ntrans_text, c.cls_hidden, c.cls_out, c.cls_layers, c.dim_ff,
                      float(c.dropout), float(c.ln_eps))


class HipError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/m2fnet_hip.h declares
SIGNATURES = {
    "m2f_last_error": (ctypes.c_char_p, []),
    "m2f_device_check": (c_int, []),
    "m2f_param_layout": (c_int, [ctypes.POINTER(M2FConfigC), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_int,
                                 ctypes.POINTER(c_int64)]),
    "m2f_workspace_bytes": (c_int64, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int]),
    "m2f_workspace_bytes_packed": (c_int64, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int, c_int]),
    "m2f_plan_create": (c_void_p, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_void_p]),
    "m2f_plan_create_packed": (c_void_p, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_void_p]),
    "m2f_param_shadow_elems": (c_int64, [ctypes.POINTER(M2FConfigC)]),
    "m2f_param_shadow_init": (c_int, [ctypes.POINTER(M2FConfigC), c_void_p, c_void_p]),
    "m2f_workspace_bytes_shared": (c_int64, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int, c_int]),
    "m2f_plan_create_shared": (c_void_p, [ctypes.POINTER(M2FConfigC), c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_void_p, c_void_p]),
    "m2f_plan_params_fresh": (c_int, [c_void_p, c_int]),
    "m2f_adam_step_shadowed": (c_int, [ctypes.POINTER(M2FConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                       c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "m2f_adam_step_shadowed_range": (c_int, [ctypes.POINTER(M2FConfigC), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "m2f_plan_skipped_copies": (c_int, [c_void_p]),
    "m2f_plan_destroy": (None, [c_void_p]),
    "m2f_plan_buffer": (c_void_p, [c_void_p, c_int]),
    "m2f_plan_num_launches": (c_int, [c_void_p, c_int]),
    "m2f_plan_persistent": (c_int, [c_void_p]),
    "m2f_gemm_ring_launches": (ctypes.c_longlong, []),
    "m2f_plan_status": (c_int, [c_void_p, ctypes.POINTER(c_uint32)]),
    "m2f_forward": (c_int, [c_void_p, c_void_p]),
    "m2f_loss": (c_int, [c_void_p, c_float, c_int, c_int, c_void_p]),
    "m2f_backward": (c_int, [c_void_p, c_void_p]),
    "m2f_step": (c_int, [c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "m2f_plan_split_offset": (c_int64, [c_void_p]),
    "m2f_step_part": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p]),
    "m2f_step_timed": (c_int, [c_void_p, c_float, c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_int),
                               ctypes.POINTER(c_float), ctypes.POINTER(ctypes.c_double)]),
    "m2f_event_overhead": (c_int, [c_void_p, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "m2f_gather_dialogues": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                     c_int, c_void_p, c_void_p, c_void_p]),
    "m2f_rng_advance": (c_int, [c_void_p, c_void_p]),
    "m2f_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_void_p, c_void_p]),
    "m2f_adam_step_g16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_void_p, c_void_p]),
    "m2f_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                         c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float,
                         c_void_p, c_int, c_int, c_int, c_int, c_uint32, c_float, c_void_p, c_int, c_void_p, c_void_p,
                         c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "m2f_attention_fwd": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int, c_void_p, c_uint32, c_float, c_void_p, c_void_p]),
    "m2f_attention_bwd": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                  c_int, c_void_p, c_int, c_uint32, c_float, c_void_p, c_void_p]),
    "m2f_attention_probs_elems": (c_int64, [c_int, c_int, c_int]),
    "m2f_set_shadow_map": (c_int, [c_void_p, c_void_p, c_int64]),
    "m2f_plan_grad_bf16": (c_int, [c_void_p, c_void_p]),
    "m2f_plan_fused_adam_setup": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "m2f_plan_fused_adam": (c_int, [c_void_p, c_int]),
    "m2f_adam_hyper": (c_int, [c_void_p, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "m2f_gemm_p8": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                            c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "m2f_gemm_fp8": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p,
                             c_int, c_int, c_void_p, c_float, c_void_p]),
    "m2f_quantize_fp8": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "m2f_embed_layernorm": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_int, c_void_p]),
    "m2f_attention_long_fwd_bf16": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "m2f_attention_long_fwd_bf16_out8": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "m2f_layernorm_fwd_out8": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float,
                                       c_void_p]),
    "m2f_layernorm_fwd_diag": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "m2f_set_shadow_only": (c_int, [c_int]),
    "m2f_attention_long_fwd": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p, c_int, c_void_p]),
    "m2f_layernorm_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                  c_void_p]),
    "m2f_layernorm_bwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "m2f_cross_entropy": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
}

_lib = None


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j4"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded shared library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback exists). "
                f"Build it with `make -C {CSRC}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        raise HipError(f"{what}: {lib().m2f_last_error().decode()} (code {code})")


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise HipError("the M2FNet HIP path needs an MI355X (gfx950) GPU; there is no CPU fallback")
    check(lib().m2f_device_check(), "m2f_device_check")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def c_param_layout(c: M2FConfig) -> Tuple[list, list, int]:
    cc = config_to_c(c)
    n_max = 4096
    offs = (c_int64 * n_max)()
    nums = (c_int64 * n_max)()
    total = c_int64(0)
    n = lib().m2f_param_layout(ctypes.byref(cc), offs, nums, n_max, ctypes.byref(total))
    if n < 0:
        raise HipError(lib().m2f_last_error().decode())
    return list(offs[:n]), list(nums[:n]), int(total.value)


def verify_layout(c: M2FConfig) -> int:
    """Cross-check layout.py against the C side; returns the flat length in elements."""
    specs, total = param_specs(c)
    uniq = [s for s in specs if not s.alias_of]
    offs, nums, ctotal = c_param_layout(c)
    if ctotal != total or offs != [s.offset for s in uniq] or nums != [s.numel for s in uniq]:
        raise HipError("flat parameter layout mismatch between layout.py and csrc/plan.hip")
    return total


class Plan:
    """One bound launch list (config, B, L, precision, train/eval) + its workspace."""

    def __init__(self, cfg: M2FConfig, B: int, L: int, precision: int, train: bool, params: torch.Tensor,
                 grads: Optional[torch.Tensor], rng_state: Optional[torch.Tensor], T: Optional[int] = None,
                 param_shadow: Optional[torch.Tensor] = None):
        """T: PACKED plan (m2f_plan_create_packed) - T token rows shared by the B dialogues through cu_seqlens; `set_inputs`
        packs the padded batch it is given and `logits` unpacks, so callers see the padded [B, L, ...] surface either way.
        param_shadow: the model's shared bf16 parameter-shadow buffer (m2f_plan_create_shared); None = the plan keeps its own."""
        require_gpu()
        self.packed = T is not None
        self.cfg, self.B, self.L, self.T = cfg, B, L, (int(T) if self.packed else B * L)
        self.precision, self.train = precision, train
        self._cc = config_to_c(cfg)
        self.shared_shadow = param_shadow is not None
        self._fresh = False
        self._on_cast = None      # engine hook: a forward that re-cast the shared parameter shadows leaves them current
        if self.shared_shadow:
            nbytes = lib().m2f_workspace_bytes_shared(ctypes.byref(self._cc), B, L, self.T if self.packed else 0, int(train))
        else:
            nbytes = (lib().m2f_workspace_bytes_packed(ctypes.byref(self._cc), B, L, self.T, int(train)) if self.packed
                      else lib().m2f_workspace_bytes(ctypes.byref(self._cc), B, L, int(train)))
        if nbytes < 0:
            raise HipError(lib().m2f_last_error().decode())
        self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=params.device)
        # the zero-fill runs on torch's current stream, m2f_plan_create uploads its tables with blocking copies on the
        # null stream: order the two whatever stream context the caller is in
        torch.cuda.current_stream(params.device).synchronize()
        base = self.workspace.data_ptr()
        self._ws_off = (-base) % 256
        self._keep = (params, grads, rng_state, param_shadow)
        if self.shared_shadow:
            self.handle = lib().m2f_plan_create_shared(ctypes.byref(self._cc), B, L, self.T if self.packed else 0, precision,
                                                       int(train), params.data_ptr(), ptr(grads), base + self._ws_off, nbytes,
                                                       ptr(rng_state), param_shadow.data_ptr())
        elif self.packed:
            self.handle = lib().m2f_plan_create_packed(ctypes.byref(self._cc), B, L, self.T, precision, int(train),
                                                       params.data_ptr(), ptr(grads), base + self._ws_off, nbytes, ptr(rng_state))
        else:
            self.handle = lib().m2f_plan_create(ctypes.byref(self._cc), B, L, precision, int(train), params.data_ptr(),
                                                ptr(grads), base + self._ws_off, nbytes, ptr(rng_state))
        if not self.handle:
            raise HipError("m2f_plan_create: " + lib().m2f_last_error().decode())
        C = cfg.cls_out
        pad8 = lambda w: (w + 7) // 8 * 8          # every activation row is padded to a multiple of 8 floats
        self.text_in = self._view(BUF_TEXT, (self.T, pad8(max(cfg.d_text, 1))), torch.float32)[:, : max(cfg.d_text, 1)]
        self.audio_in = self._view(BUF_AUDIO, (self.T, pad8(max(cfg.d_audio, 1))), torch.float32)[:, : max(cfg.d_audio, 1)]
        self.keypad_in = self._view(BUF_KEYPAD, (self.T,), torch.uint8)
        self.labels_in = self._view(BUF_LABELS, (self.T,), torch.int64)
        self.class_w = self._view(BUF_CLASSW, (16,), torch.float32)
        self._logits = self._view(BUF_LOGITS, (self.T, C) if self.packed else (B, L, C), torch.float32)
        self.cu_in = self._view(BUF_CU_SEQLENS, (B + 1,), torch.int32)
        self._dst = self._valid = None        # packed plans: token row of every (dialogue, slot) of the last batch; its validity
        if train and grads is not None:
            # (loss, den, num) live in the tail of the flat gradient buffer (see include/m2fnet_hip.h)
            assert grads.numel() >= params.numel() + 4, "gradient buffer needs a 64-float tail"
            self.loss = grads[params.numel(): params.numel() + 4]
            assert self.loss.data_ptr() == lib().m2f_plan_buffer(self.handle, BUF_LOSS)
        else:
            self.loss = self._view(BUF_LOSS, (4,), torch.float32)
        self._dlogits = self._view(BUF_DLOGITS, (self.T, C) if self.packed else (B, L, C), torch.float32)
        self._fam0_out = (self._view(BUF_FAM0_OUT, (self.T, pad8(cfg.d_fam)) if self.packed else (B, L, pad8(cfg.d_fam)),
                                     torch.float32)[..., : cfg.d_fam] if cfg.fam_enabled else None)
        # shape of the batch last handed to set_inputs: a plan may be larger than the batch it runs (shape buckets), the
        # result views below are cut to the batch
        self.in_B, self.in_L = B, L
        self.version = 0          # bumped by every forward; backward checks it still owns the activations
        self._pending = None      # weak reference to the autograd node whose backward still needs this plan's activations

    def _unpack(self, rows: torch.Tensor) -> torch.Tensor:
        """[T, C] rows of a packed plan -> the padded [b, l, C] surface of the last batch (pad slots = 0)."""
        return rows[self._dst] * self._valid[..., None].to(rows.dtype)

    @property
    def logits(self) -> torch.Tensor:
        if self.packed:
            return self._unpack(self._logits)
        return self._logits[: self.in_B, : self.in_L]

    @property
    def dlogits(self) -> torch.Tensor:
        if self.packed:
            return self._unpack(self._dlogits)
        return self._dlogits[: self.in_B, : self.in_L]

    @property
    def fam0_out(self) -> Optional[torch.Tensor]:
        if self._fam0_out is None:
            return None
        return self._unpack(self._fam0_out) if self.packed else self._fam0_out[: self.in_B, : self.in_L]

    def _h(self) -> int:
        """The C handle; raises (instead of handing NULL to the library) once the plan was closed."""
        if not self.handle:
            raise HipError("this plan was closed (evicted from the engine's plan cache or destroyed): its workspace and launch "
                           "lists are gone")
        return self.handle

    def _view(self, which: int, shape, dtype) -> torch.Tensor:
        p = lib().m2f_plan_buffer(self.handle, which)
        if not p:
            raise HipError(f"plan buffer {which} missing")
        off = p - self.workspace.data_ptr()
        n = 1
        for s in shape:
            n *= s
        esize = torch.empty(0, dtype=dtype).element_size()
        return self.workspace[off: off + n * esize].view(dtype).view(*shape)

    def persistent(self) -> int:
        """Always 0 since round 4 (the persistent strip-dataflow kernels were removed; kept for callers that still ask)."""
        return lib().m2f_plan_persistent(self._h())

    def check_status(self) -> None:
        """Raises on a destroyed plan; nothing else can be wrong since the persistent kernels (bounded waits) were removed."""
        out = (c_uint32 * 8)()
        check(lib().m2f_plan_status(self._h(), out), "m2f_plan_status")

    def num_launches(self) -> Dict[str, int]:
        return {k: lib().m2f_plan_num_launches(self._h(), i) for i, k in enumerate(("forward", "loss", "backward"))}

    def params_fresh(self, fresh: bool) -> None:
        """Declare the shared parameter shadows current (the optimizer wrote them) or stale (the forward re-casts them)."""
        fresh = bool(fresh) and self.shared_shadow
        if fresh != self._fresh:
            check(lib().m2f_plan_params_fresh(self._h(), int(fresh)), "m2f_plan_params_fresh")
            self._fresh = fresh

    def hold(self, node) -> None:
        """A forward ran under autograd: `node` (its grad_fn) will call backward() on these activations."""
        import weakref
        self._pending = weakref.ref(node) if node is not None else None

    def release(self) -> None:
        self._pending = None

    def busy(self) -> bool:
        """True while an autograd graph that has not run its backward yet (and is still alive) owns the activations."""
        return self._pending is not None and self._pending() is not None

    def set_inputs(self, text: Optional[torch.Tensor], audio: Optional[torch.Tensor], key_pad: torch.Tensor,
                   labels: Optional[torch.Tensor] = None) -> None:
        """Device-to-device copies of one batch into the plan's staging buffers (async on the stream).

        The batch may be SMALLER than the plan (b <= B dialogues of l <= L utterances: the engine rounds shapes up to a few
        buckets so that the variable dialogue lengths of real data do not create a plan per length).  The extra slots are
        padding in the reference's own sense - zero features, padding_mask = True, label -1 (src/utils.py:15-31) - and each
        extra DIALOGUE keeps one unmasked, unlabeled slot (a fully masked dialogue would produce NaN logits in the reference
        too): it flows through the forward, contributes exactly zero to the loss and to every gradient,
        and its logits are cut off by the `logits` view.  Valid logits do not depend on padding (SURVEY 8-a fact i)."""
        self._h()
        b, l = key_pad.shape if key_pad.dim() == 2 else (self.B, self.L)
        if b > self.B or l > self.L:
            raise HipError(f"batch {b} x {l} does not fit the plan {self.B} x {self.L}")
        self.in_B, self.in_L = b, l
        if self.packed:
            return self._set_inputs_packed(text, audio, key_pad.reshape(b, l), labels)
        if (b, l) == (self.B, self.L):
            if text is not None and self.cfg.text_enabled:
                self.text_in.copy_(text.reshape(self.T, -1), non_blocking=True)
            if audio is not None and self.cfg.audio_enabled:
                self.audio_in.copy_(audio.reshape(self.T, -1), non_blocking=True)
            self.keypad_in.copy_(key_pad.reshape(self.T), non_blocking=True)
            if labels is not None:
                self.labels_in.copy_(labels.reshape(self.T), non_blocking=True)
            return
        B, L = self.B, self.L
        if text is not None and self.cfg.text_enabled:
            self.text_in.zero_()
            self.text_in.view(B, L, -1)[:b, :l].copy_(text, non_blocking=True)
        if audio is not None and self.cfg.audio_enabled:
            self.audio_in.zero_()
            self.audio_in.view(B, L, -1)[:b, :l].copy_(audio, non_blocking=True)
        kp = self.keypad_in.view(B, L)
        kp.fill_(1)
        kp[:b, :l].copy_(key_pad, non_blocking=True)
        if b < B:
            kp[b:, 0] = 0                           # one live (unlabeled) slot per filler dialogue
        self.labels_in.fill_(-1)
        if labels is not None:
            self.labels_in.view(B, L)[:b, :l].copy_(labels, non_blocking=True)

    def _set_inputs_packed(self, text, audio, key_pad, labels) -> None:
        """Packs a padded batch: the valid slots of dialogue b (in order) become token rows cu[b] .. cu[b+1]-1, no sync with
        the host.  Filler dialogues of a bucketed plan get one zero, unlabeled row each; row T-1 absorbs the scatter of the
        pad slots and is zeroed afterwards (the engine sizes T for valid + fillers + 1 rows).  Rows past the last dialogue are
        padding: zero features, label -1 - they contribute exact zeros to the loss and to every gradient."""
        b, l = key_pad.shape
        dev, T = key_pad.device, self.T
        valid = ~key_pad.bool()
        rank = torch.cumsum(valid, 1, dtype=torch.int64) - 1                     # position among the dialogue's valid slots
        lens = valid.sum(1, dtype=torch.int64)
        cu = torch.zeros(self.B + 1, dtype=torch.int64, device=dev)
        cu[1: b + 1] = torch.cumsum(lens, 0)
        if b < self.B:
            cu[b + 1:] = cu[b] + torch.arange(1, self.B - b + 1, device=dev)
        self.cu_in.copy_(cu.to(torch.int32), non_blocking=True)
        dst = torch.where(valid, cu[:b, None] + rank, torch.full_like(rank, T - 1))
        self._dst, self._valid = dst, valid
        flat = dst.reshape(-1)
        for buf, src, on in ((self.text_in, text, self.cfg.text_enabled), (self.audio_in, audio, self.cfg.audio_enabled)):
            if src is not None and on:
                buf.zero_()
                buf.index_copy_(0, flat, src.reshape(b * l, -1).to(buf.dtype))
                buf[T - 1].zero_()
        self.keypad_in.zero_()
        self.labels_in.fill_(-1)
        if labels is not None:
            self.labels_in.index_copy_(0, flat, labels.reshape(-1).to(torch.int64))
            self.labels_in[T - 1] = -1

    def set_dlogits(self, g: torch.Tensor) -> None:
        """d loss / d logits of the last batch ([b, l, C], padded surface) into the plan's buffer."""
        if self.packed:
            self._dlogits.zero_()
            self._dlogits.index_copy_(0, self._dst.reshape(-1), (g * self._valid[..., None].to(g.dtype)).reshape(-1, g.shape[-1]))
            self._dlogits[self.T - 1].zero_()
            return
        if self.in_B != self.B or self.in_L != self.L:
            self._dlogits.zero_()                 # filler slots of a bucketed plan carry no gradient
        self._dlogits[: self.in_B, : self.in_L].copy_(g.reshape(self.in_B, self.in_L, -1))

    def _casted(self) -> None:
        if self.shared_shadow and not self._fresh and self._on_cast is not None:
            self._on_cast()

    def forward(self) -> torch.Tensor:
        self.version += 1
        check(lib().m2f_forward(self._h(), stream_ptr()), "m2f_forward")
        self._casted()
        return self.logits

    def nbytes(self) -> int:
        return self.workspace.numel()

    def loss_fwd(self, label_smoothing: float = 0.1, use_class_weights: bool = False, normalise: bool = True):
        check(lib().m2f_loss(self._h(), label_smoothing, int(use_class_weights), int(normalise), stream_ptr()),
              "m2f_loss")
        return self.loss

    def backward(self) -> None:
        check(lib().m2f_backward(self._h(), stream_ptr()), "m2f_backward")

    def step(self, label_smoothing: float = 0.1, use_class_weights: bool = False, normalise: bool = True,
             use_graph: bool = True) -> torch.Tensor:
        self.version += 1
        check(lib().m2f_step(self._h(), label_smoothing, int(use_class_weights), int(normalise), int(use_graph),
                             stream_ptr()), "m2f_step")
        self._casted()
        return self.loss

    def fused_adam_setup(self, params, exp_avg, exp_avg_sq, param_shadow, hyper, grad_scale=None) -> None:
        """m2f_plan_fused_adam_setup: the optimizer's buffers for steps that apply Adam inside the weight-gradient launch (raises when
        the plan cannot: fp32 mode, another table form, per-plan shadows)."""
        check(lib().m2f_plan_fused_adam_setup(self._h(), params.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param_shadow.data_ptr(),
                                              hyper.data_ptr(), ptr(grad_scale)), "m2f_plan_fused_adam_setup")
        self._fused_refs = (params, exp_avg, exp_avg_sq, param_shadow, hyper, grad_scale)      # (the plan holds raw pointers)

    def grad_bf16(self, buf16) -> None:
        """m2f_plan_grad_bf16: the NEXT steps leave every gradient, rounded once, in `buf16` (bf16 [n_params]; None: back to fp32)."""
        check(lib().m2f_plan_grad_bf16(self._h(), buf16.data_ptr() if buf16 is not None else None), "m2f_plan_grad_bf16")
        self._g16_ref = buf16

    def fused_adam(self, on: bool) -> None:
        """The NEXT step() also takes the optimizer step (on) / leaves the weight gradients in the gradient buffer (off)."""
        check(lib().m2f_plan_fused_adam(self._h(), int(bool(on))), "m2f_plan_fused_adam")

    def skipped_copies(self) -> int:
        """How many fp32 / bf16 copies of activations this plan does not write because nobody reads them (bf16 mode)."""
        return int(lib().m2f_plan_skipped_copies(self._h()))

    def split_offset(self) -> int:
        """First element of the flat gradient buffer that is final after `step_part(0)` (0: this plan cannot be split)."""
        return int(lib().m2f_plan_split_offset(self._h()))

    def step_part(self, part: int, label_smoothing: float = 0.1, use_class_weights: bool = False, normalise: bool = True,
                  use_graph: bool = True) -> torch.Tensor:
        """m2f_step_part: part 0 = forward + criterion + classifier / fusion backward (+ their weight gradients), part 1 = the rest."""
        if part == 0:
            self.version += 1
        check(lib().m2f_step_part(self._h(), int(part), label_smoothing, int(use_class_weights), int(normalise), int(use_graph),
                                  stream_ptr()), "m2f_step_part")
        if part == 0:
            self._casted()
        return self.loss

    def step_timed(self, label_smoothing: float = 0.1, use_class_weights: bool = False, normalise: bool = True):
        """One eager step with per-launch hipEvent timing -> list of (kind, ms, algorithmic flops)."""
        self.version += 1
        n_max = 4096
        kinds, ms, fl = (c_int * n_max)(), (c_float * n_max)(), (ctypes.c_double * n_max)()
        n = lib().m2f_step_timed(self._h(), label_smoothing, int(use_class_weights), int(normalise), stream_ptr(),
                                 n_max, kinds, ms, fl)
        if n < 0:
            raise HipError("m2f_step_timed: " + lib().m2f_last_error().decode())
        self._casted()
        return [(kinds[i], ms[i], fl[i]) for i in range(n)]

    def close(self) -> None:
        """Destroy the plan (captured graph, launch lists) and drop its workspace."""
        h = getattr(self, "handle", None)
        if h and _lib is not None:
            _lib.m2f_plan_destroy(h)
        self.handle = None
        self.workspace = None

    def __del__(self):
        self.close()


def event_overhead(pairs: int = 200):
    """-> (empty hipEvent pair, pair around a one-thread kernel) in ms: what a step_timed interval holds besides the kernel."""
    scratch = torch.zeros(4, dtype=torch.int32, device="cuda")
    a, b = c_float(0), c_float(0)
    check(lib().m2f_event_overhead(scratch.data_ptr(), pairs, ctypes.byref(a), ctypes.byref(b), stream_ptr()), "m2f_event_overhead")
    return a.value, b.value


def param_shadow_buffer(cfg: M2FConfig, device) -> torch.Tensor:
    """The shared bf16 parameter-shadow buffer of a model (+ the fused optimizer's tensor table behind it), initialised."""
    cc = config_to_c(cfg)
    n = lib().m2f_param_shadow_elems(ctypes.byref(cc))
    if n < 0:
        raise HipError(lib().m2f_last_error().decode())
    buf = torch.empty(n + 128, dtype=torch.int16, device=device)
    off = ((-buf.data_ptr()) % 256) // 2
    buf = buf[off: off + n]
    torch.cuda.current_stream(buf.device).synchronize()
    check(lib().m2f_param_shadow_init(ctypes.byref(cc), buf.data_ptr(), stream_ptr()), "m2f_param_shadow_init")
    return buf


def adam_step_shadowed(cfg: M2FConfig, params, grads, exp_avg, exp_avg_sq, param_shadow, step: int, lr: float,
                       betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                       grad_scale: Optional[torch.Tensor] = None, first: int = 0, end: int = -1) -> None:
    """torch.optim.Adam's update over the flat buffers + the bf16 shadows of every 2-D parameter (m2f_adam_step_shadowed_range).
    The buffers are always passed WHOLE; `[first, end)` - offsets of parameter tensors, end < 0: to the last one - selects what is
    updated; `grads` fp32, or bf16 (the reduced buffer of the data-parallel bf16 exchange, same indexing)."""
    cc = config_to_c(cfg)
    check(lib().m2f_adam_step_shadowed_range(ctypes.byref(cc), params.data_ptr(), grads.data_ptr(), int(grads.dtype == torch.bfloat16),
                                             exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param_shadow.data_ptr(), int(first), int(end),
                                             lr, betas[0], betas[1], eps, weight_decay, step, ptr(grad_scale), stream_ptr()),
          "m2f_adam_step_shadowed_range")


def adam_hyper(hyper: torch.Tensor, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """The step-dependent factors of Adam's update into 8 device floats (read by the fused step's kernels), on the current stream."""
    check(lib().m2f_adam_hyper(hyper.data_ptr(), lr, betas[0], betas[1], eps, weight_decay, int(step), stream_ptr()), "m2f_adam_hyper")


def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
              lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
              grad_scale: Optional[torch.Tensor] = None) -> None:
    """grads: fp32, or bf16 (the reduced buffer of the data-parallel bf16 exchange) - same update, fp32 state either way."""
    if grads.dtype == torch.bfloat16:
        check(lib().m2f_adam_step_g16(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                      params.numel(), lr, betas[0], betas[1], eps, weight_decay, step, ptr(grad_scale),
                                      stream_ptr()), "m2f_adam_step_g16")
        return
    check(lib().m2f_adam_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                              params.numel(), lr, betas[0], betas[1], eps, weight_decay, step, ptr(grad_scale),
                              stream_ptr()), "m2f_adam_step")
