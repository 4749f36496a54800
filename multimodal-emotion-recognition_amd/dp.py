"""Dialogue-sharded data parallelism (the reference has none: single process, src/train.py:20).

Dialogues are independent units of work (SURVEY.md 8-a fact ii), so each rank (one process per GPU) runs the
full step on its own dialogues and the only exchange is ONE sum-all-reduce of the flat gradient buffer over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" on CPU for the tests).

Loss normalisation.  The criterion is a mean over the VALID utterances of the global batch
(CrossEntropyLoss(ignore_index=-1), src/train.py:48-50).  Averaging per-rank means is wrong when ranks hold
different valid counts, so each rank back-propagates the gradient of its SUM of per-utterance terms
(``m2f_step(normalise=0)``); the criterion kernel writes (denominator, numerator) into the tail of the flat
gradient buffer; after
the all-reduce every rank holds  sum_r g_r  and  D = sum_r den_r , and the fused Adam kernel divides by D on the
device (no host sync).  loss = sum_r num_r / D.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TAIL = 64           # floats appended to the flat gradient buffer: [loss, den, num, 0, ...] written by the criterion kernel


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ        # under torchrun (also with one rank)
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:                              # M2F_DIST_BACKEND=gloo: several ranks sharing ONE GPU (rehearsals / tests)
            backend = os.environ.get("M2F_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        # RCCL prints a version banner on STDOUT when the communicator is created (lazily, at the first collective);
        # callers such as bench.py own stdout (one JSON line), so create it now with stdout pointed at stderr.
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend, rank=rank, world_size=world)
            probe = torch.zeros(1, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if backend == "nccl":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return rank, world, local


def shard_dialogues(n_dialogues: int, rank: int, world: int) -> List[int]:
    """Rank r owns dialogues r, r+W, r+2W, ... of the global batch (no data-path collective)."""
    return list(range(rank, n_dialogues, world))


class ShardedLoader:
    """Every rank iterates the SAME global batches (same loader, same shuffle seed) and keeps its dialogues r, r+W, ... of
    each: the global batch - and with it the optimisation trajectory - is what the single-process run of the reference
    (src/train.py:26-33) would see.  A batch with fewer dialogues than ranks leaves some ranks an EMPTY shard (B = 0): the
    step still has to be taken (DataParallelStep adds a zero contribution) or the other ranks would wait in the all-reduce."""

    def __init__(self, loader, rank: int, world: int):
        self.loader, self.rank, self.world = loader, rank, world

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            n = batch["padding_mask"].shape[0]
            mine = shard_dialogues(n, self.rank, self.world)
            if mine:
                idx = torch.as_tensor(mine, device=batch["padding_mask"].device)
                out = {k: v.index_select(0, idx) for k, v in batch.items()}
                # drop the columns that are padding for every dialogue of the shard (the batch was padded to ITS longest)
                keep = int((~out["padding_mask"]).sum(dim=1).max().item())
                out = {k: v[:, :keep].contiguous() for k, v in out.items()}
            else:
                out = {k: v[:0] for k, v in batch.items()}
            yield out


def broadcast_from_rank0(values: Sequence[float], device=None) -> List[float]:
    """Rank 0's values on every rank (validation metrics, so that early stopping takes the same decision everywhere)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if t.device.type == "cpu" and dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
    return t.cpu().tolist()


class _StagedWork:
    """gloo has no device collectives on this build: a CUDA tensor is summed through a host copy.  Same interface as the
    Work object of an asynchronous collective (only `wait`), same stream semantics as RCCL's: after `wait()` the current
    stream may read the result."""

    def __init__(self, t: torch.Tensor, group):
        self.t, self.group = t, group

    def wait(self) -> None:
        host = self.t.detach().to("cpu", torch.float32)        # (synchronises with the current stream: the producers are done)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        self.t.copy_(host.to(self.t.dtype))


def _all_reduce_sum(t: torch.Tensor, group=None):
    """Asynchronous SUM all-reduce of `t` -> an object with wait()."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return _StagedWork(t, group)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


class GradReducer:
    """Sum-all-reduce of [flat gradients | den | num] in `n_buckets` contiguous chunks.

    ``buf`` is the rank's flat gradient buffer extended by TAIL floats.  Chunks are issued in order on the
    collective's own stream (``async_op=True``), so a caller that produces gradients back-to-front can start
    reducing finished chunks while the rest is still being computed."""

    def __init__(self, buf: torch.Tensor, n_params: int, group=None, n_buckets: int = 1, exchange: str = "fp32"):
        """exchange = "fp32": the buffer itself is all-reduced (exact sum of the ranks' fp32 gradients).
        exchange = "bf16": the parameter gradients travel as bf16 (half the xGMI bytes; the 64-float tail with the
        loss terms stays fp32 in its own small collective) and the optimizer reads the reduced bf16 buffer directly -
        the usual mixed-precision trade (each rank's gradient rounded once to bf16, ring partial sums in bf16), offered
        for the bf16 compute mode only."""
        assert buf.numel() >= n_params + 3 and buf.dim() == 1
        assert exchange in ("fp32", "bf16")
        self.buf, self.n, self.group, self.exchange = buf, n_params, group, exchange
        n_buckets = max(1, int(n_buckets))
        edges = [round(i * buf.numel() / n_buckets / 64) * 64 for i in range(n_buckets)] + [buf.numel()]
        self.chunks = [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]
        # bf16 exchange: buckets over the parameter part only
        pedges = [round(i * n_params / n_buckets / 64) * 64 for i in range(n_buckets)] + [n_params]
        self.param_chunks = [(a, b) for a, b in zip(pedges[:-1], pedges[1:]) if b > a]
        if exchange == "bf16" and dist.is_initialized():
            # probe once, on every rank alike: a backend without bf16 reductions falls back to the exact fp32 exchange
            try:
                probe = torch.zeros(64, dtype=torch.bfloat16, device=buf.device)
                _all_reduce_sum(probe, group).wait()
            except (RuntimeError, ValueError, TypeError) as e:          # pragma: no cover - depends on the backend build
                import sys
                print(f"mer_amd.dp: bf16 all-reduce unavailable ({e}); using the fp32 gradient exchange", file=sys.stderr)
                self.exchange = exchange = "fp32"
        self.buf16 = torch.empty(n_params, dtype=torch.bfloat16, device=buf.device) if exchange == "bf16" else None
        self._work = []
        self._starts = None

    def align_to(self, tensor_starts) -> None:
        """Move the bucket edges to the nearest parameter-tensor boundaries (offsets into the flat buffer): the fused optimizer can
        then step a bucket with its shadow-writing kernel (optim.FusedAdam.step_ranges), which works on whole tensors."""
        starts = sorted(set(int(o) for o in tensor_starts))
        if not starts or starts[0] != 0:
            return
        self._starts = starts

        def snap_all(chunks, last):
            edges = sorted(set([0] + [self._snap(a) for a, _ in chunks[1:]] + [last]))
            return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]
        self.chunks = snap_all(self.chunks, self.buf.numel())
        self.param_chunks = snap_all(self.param_chunks, self.n)

    def _snap(self, x: int) -> int:
        if not self._starts:
            return x
        import bisect
        i = bisect.bisect_left(self._starts, x)
        cands = [self._starts[j] for j in (i - 1, i) if 0 <= j < len(self._starts)]
        return min(cands, key=lambda c: abs(c - x))

    @property
    def tail(self) -> torch.Tensor:
        return self.buf[self.n: self.n + 3]              # (local loss, den, num); den and num are summed

    def set_loss_terms(self, den: torch.Tensor, num: torch.Tensor) -> None:
        """Only for buffers not filled by a train plan (the plan's criterion kernel writes the tail itself)."""
        self.buf[self.n + 1: self.n + 2].copy_(den.reshape(1))
        self.buf[self.n + 2: self.n + 3].copy_(num.reshape(1))

    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def zero_contribution(self) -> None:
        """This rank holds no dialogue of the global batch: gradients, denominator and numerator are all zero."""
        self.buf.zero_()

    def all_reduce(self, async_op: bool = False) -> None:
        if not dist.is_initialized():
            return                                   # single process: nothing to exchange
        self._work = [_all_reduce_sum(self.buf[a:b], self.group) for a, b in self.chunks]
        if not async_op:
            self.wait()

    def wait(self) -> None:
        for w in self._work:
            w.wait()
        self._work = []

    def reduce_and_step(self, optimizer) -> None:
        """Pipelined exchange + update: the buckets are all-reduced LAST FIRST (the tail bucket carries the global
        denominator every update needs), and the fused-Adam launch of bucket k only waits for bucket k's collective,
        so the HBM-bound optimizer runs under the xGMI-bound exchange of the following buckets."""
        if not dist.is_initialized():
            optimizer.step()
            return
        if self.exchange == "bf16":
            tail = self.buf[self.n:]
            work_tail = _all_reduce_sum(tail, self.group)   # (loss, den, num): fp32
            order = list(reversed(self.param_chunks))
            work = []
            for a, b in order:
                self.buf16[a:b].copy_(self.buf[a:b])                     # one rounding per rank, on the device
                work.append(_all_reduce_sum(self.buf16[a:b], self.group))

            def wait(i):
                if i == 0:
                    work_tail.wait()
                work[i].wait()
            optimizer.step_ranges(order, before_each=wait, grads=self.buf16)
            return
        order = list(reversed(self.chunks))
        work = [_all_reduce_sum(self.buf[a:b], self.group) for a, b in order]
        optimizer.step_ranges(order, before_each=lambda i: work[i].wait())

    def reduce_and_step_split(self, optimizer, run_rest, split: int) -> None:
        """Exchange overlapped with the backward pass: the caller has run part 0 of the step (everything from element `split` of
        the buffer on is final: the fusion stack's and the classifier's gradients + the loss tail), `run_rest()` enqueues part 1
        (the encoders' backward).  The tail bucket's all-reduce is put on the wire FIRST - torch's asynchronous collectives wait for
        the work already queued on the current stream, i.e. for part 0 only - then part 1 is enqueued and computes while that
        bucket travels; the encoder buckets follow last-first as in `reduce_and_step`, each bucket's fused-Adam launch behind its
        own collective.  At C3 the tail bucket is 35.5 M of 103.8 M parameters (71 MB of the 198 MB bf16 exchange)."""
        if not dist.is_initialized():
            run_rest()
            optimizer.step()
            return
        split = max(0, min(int(split) // 64 * 64, self.n))
        n_b = max(1, len(self.chunks) - 1)
        edges = sorted(set([self._snap(round(i * split / n_b / 64) * 64) for i in range(n_b)] + [split]))
        head = [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]            # encoder part of the parameters, in buckets
        if self.exchange == "bf16":
            tail = self.buf[self.n:]
            work_tail = _all_reduce_sum(tail, self.group)                       # (loss, den, num): fp32
            self.buf16[split:self.n].copy_(self.buf[split:self.n])
            work_first = _all_reduce_sum(self.buf16[split:self.n], self.group) if self.n > split else None
            run_rest()
            order = [(split, self.n)] + list(reversed(head))
            work = [work_first]
            for a, b in order[1:]:
                self.buf16[a:b].copy_(self.buf[a:b])
                work.append(_all_reduce_sum(self.buf16[a:b], self.group))

            def wait(i):
                if i == 0:
                    work_tail.wait()
                if work[i] is not None:
                    work[i].wait()
            optimizer.step_ranges(order, before_each=wait, grads=self.buf16)
            return
        work_first = _all_reduce_sum(self.buf[split:], self.group)               # fusion stack + classifier + (loss, den, num)
        run_rest()
        order = [(split, self.buf.numel())] + list(reversed(head))
        work = [work_first] + [_all_reduce_sum(self.buf[a:b], self.group) for a, b in order[1:]]
        optimizer.step_ranges(order, before_each=lambda i: work[i].wait())

    @property
    def global_den(self) -> torch.Tensor:          # device scalar view, feeds m2f_adam_step(grad_scale_ptr)
        return self.buf[self.n + 1: self.n + 2]

    def global_loss(self) -> torch.Tensor:
        return self.buf[self.n + 2] / self.buf[self.n + 1]


def sum_over_ranks(values: Sequence[float], device=None) -> List[float]:
    """SUM over ranks, float64 (per-batch validation sums: every rank ends with the same totals)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if t.device.type == "cpu" and dist.get_backend() == "nccl":
            t = t.cuda()
        elif t.device.type == "cuda" and dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


def dropout_seed(initial_seed: int, rank: int) -> Tuple[int, int]:
    """(lo, hi) 32-bit words of the dropout generator's seed on `rank`: the process seed with the rank folded in, so that the
    replicas (same torch seed = same initial weights) draw DIFFERENT masks for their different dialogues - rank 0 keeps the
    single-process stream."""
    seed = (int(initial_seed) ^ (int(rank) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF


def reduce_metrics(values: Sequence[float], device=None) -> List[float]:
    """MAX over ranks (timings) - used by bench.py."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if t.device.type == "cpu" and dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().tolist()


class DataParallelStep:
    """One optimizer step of dialogue-sharded data-parallel training:
    m2f_step(normalise=0) -> tail <- (den, num) -> all-reduce -> fused Adam with grad_scale = global den."""

    def __init__(self, model, optimizer, group=None, n_buckets: int = 4, exchange: str = "fp32", overlap: bool = True):
        """overlap: with more than one rank, run the step in two parts (runtime.Plan.step_part) and put the all-reduce of the
        fusion stack's / classifier's gradients on the wire before the encoders' backward starts (plans that cannot be split -
        fp32 mode - exchange after the whole backward as before)."""
        self.model, self.optimizer, self.overlap = model, optimizer, overlap
        eng = model.engine()
        eng.ensure_grad()
        self.reducer = GradReducer(eng.flat_grad_ext, eng.flat.numel(), group, n_buckets, exchange)
        self.reducer.align_to(o for (_, o, _, _) in eng.items)      # buckets of whole tensors: the optimizer keeps the bf16 parameter shadows current
        optimizer.grad_scale = self.reducer.global_den

    def __call__(self, text, audio, mask, emotion, label_smoothing: float = 0.1, class_weights=None,
                 use_graph: bool = True) -> torch.Tensor:
        eng = self.model.engine()
        B, L = mask.shape
        cur = torch.cuda.current_stream(eng.device)
        eng.stream.wait_stream(cur)
        if B == 0:                                    # empty shard: contribute zeros, but take part in every collective
            with torch.cuda.stream(eng.stream):
                self.reducer.zero_contribution()
                eng.publish_grads()
                self.reducer.reduce_and_step(self.optimizer)
                loss = self.reducer.global_loss()
            cur.wait_stream(eng.stream)
            return loss
        plan = eng.plan(B, L, True, self.model.training and self.model.m2f_config.dropout > 0.0)
        with torch.cuda.stream(eng.stream):
            plan.set_inputs(text if self.model.text_enabled else None, audio if self.model.audio_enabled else None,
                            mask, emotion)
            if class_weights is not None:
                plan.class_w[: class_weights.numel()].copy_(class_weights)
            split = plan.split_offset() if (self.overlap and self.reducer.world() > 1) else 0
            if split > 0:
                cw = class_weights is not None
                plan.step_part(0, label_smoothing, cw, False, use_graph)              # tail <- (loss, den, num); fusion / classifier gradients final
                eng.publish_grads()
                self.reducer.reduce_and_step_split(self.optimizer, lambda: plan.step_part(1, label_smoothing, cw, False, use_graph), split)
            else:
                plan.step(label_smoothing, class_weights is not None, False, use_graph)   # tail <- (loss, den, num)
                eng.publish_grads()
                self.reducer.reduce_and_step(self.optimizer)
            loss = self.reducer.global_loss()
        cur.wait_stream(eng.stream)
        return loss
