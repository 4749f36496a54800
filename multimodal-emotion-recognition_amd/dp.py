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


ALGORITHMS = ("all_reduce", "rs_ag")


def _rs_ag_host(host: torch.Tensor, group) -> None:
    """reduce-scatter + all-gather of a host tensor whose length divides by the world size (in place)."""
    w = dist.get_world_size(group)
    shard = torch.empty(host.numel() // w, dtype=host.dtype)
    dist.reduce_scatter_tensor(shard, host, op=dist.ReduceOp.SUM, group=group)
    dist.all_gather_into_tensor(host, shard, group=group)


class _StagedWork:
    """gloo has no device collectives on this build: a CUDA tensor is summed through a host copy.  Same interface as the
    Work object of an asynchronous collective (only `wait`), same stream semantics as RCCL's: after `wait()` the current
    stream may read the result."""

    def __init__(self, t: torch.Tensor, group, algorithm: str = "all_reduce"):
        self.t, self.group, self.algorithm = t, group, algorithm

    def wait(self) -> None:
        host = self.t.detach().to("cpu", torch.float32)        # (synchronises with the current stream: the producers are done)
        if self.algorithm == "rs_ag" and host.numel() % dist.get_world_size(self.group) == 0:
            _rs_ag_host(host, self.group)
        else:
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        self.t.copy_(host.to(self.t.dtype))


class _HostPairWork:
    def __init__(self, t: torch.Tensor, group):
        self.t, self.group = t, group

    def wait(self) -> None:
        if self.t.numel() % dist.get_world_size(self.group) == 0 and self.t.is_contiguous():
            _rs_ag_host(self.t, self.group)
        else:
            dist.all_reduce(self.t, op=dist.ReduceOp.SUM, group=self.group)


class _NoWork:
    def wait(self) -> None:
        pass


class _PairWork:
    """reduce-scatter then all-gather, both asynchronous on the collective's stream (which runs them in issue order)."""

    def __init__(self, works, keep):
        self.works, self.keep = works, keep                # `keep`: the shard buffer must outlive the collectives

    def wait(self) -> None:
        for w in self.works:
            w.wait()
        self.keep = None


def _all_reduce_sum(t: torch.Tensor, group=None, algorithm: str = "all_reduce"):
    """Asynchronous SUM all-reduce of `t` -> an object with wait().

    algorithm = "rs_ag" (SURVEY section 5's direct form): ``reduce_scatter_tensor`` into this rank's 1/W shard followed by
    ``all_gather_into_tensor`` back into `t` - on 8 fully connected GPUs every rank exchanges 1/W of the bucket with each of
    its 7 peers, one peer per xGMI link, instead of passing 2 (W-1)/W of it around a ring.  Same sums as the all-reduce for
    two ranks (a + b), the same up to the order of the partial sums beyond.  Buckets whose length does not divide by W (never
    the case for W = 2, 4, 8: bucket edges are 64-float aligned) keep the plain all-reduce."""
    if dist.get_backend(group) == "gloo":
        if t.is_cuda:
            return _StagedWork(t, group, algorithm)
        if algorithm == "rs_ag":                   # gloo runs asynchronous collectives in no particular order: pair them at wait()
            return _HostPairWork(t, group)
    w = dist.get_world_size(group)
    if algorithm == "rs_ag" and w > 1 and t.numel() % w == 0 and t.is_contiguous():
        shard = torch.empty(t.numel() // w, dtype=t.dtype, device=t.device)
        w1 = dist.reduce_scatter_tensor(shard, t, op=dist.ReduceOp.SUM, group=group, async_op=True)
        w2 = dist.all_gather_into_tensor(t, shard, group=group, async_op=True)
        return _PairWork([w1, w2], shard)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


class GradReducer:
    """Sum-all-reduce of [flat gradients | den | num] in `n_buckets` contiguous chunks.

    ``buf`` is the rank's flat gradient buffer extended by TAIL floats.  Chunks are issued in order on the
    collective's own stream (``async_op=True``), so a caller that produces gradients back-to-front can start
    reducing finished chunks while the rest is still being computed."""

    def __init__(self, buf: torch.Tensor, n_params: int, group=None, n_buckets: int = 1, exchange: str = "fp32",
                 algorithm: str = "all_reduce"):
        """algorithm = "all_reduce" | "rs_ag" (see `_all_reduce_sum`).
        exchange = "fp32": the buffer itself is all-reduced (exact sum of the ranks' fp32 gradients).
        exchange = "bf16": the parameter gradients travel as bf16 (half the xGMI bytes; the 64-float tail with the
        loss terms stays fp32 in its own small collective) and the optimizer reads the reduced bf16 buffer directly -
        the usual mixed-precision trade (each rank's gradient rounded once to bf16, ring partial sums in bf16), offered
        for the bf16 compute mode only."""
        assert buf.numel() >= n_params + 3 and buf.dim() == 1
        assert exchange in ("fp32", "bf16") and algorithm in ALGORITHMS
        self.buf, self.n, self.group, self.exchange, self.algorithm = buf, n_params, group, exchange, algorithm
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
        self.buf16_filled = False          # set by the caller per step: the plan wrote buf16 itself (runtime.Plan.grad_bf16), no rounding pass
        self._work = []
        self._starts = None
        self.stub = False

    def _sum(self, t: torch.Tensor, plain: bool = False):
        """One bucket's collective (`plain`: the 64-float tail, always a plain all-reduce).  `self.stub` (bench.py's
        exposed-communication measurement): the same step with every collective replaced by a no-op on identical data."""
        if self.stub:
            return _NoWork()
        return _all_reduce_sum(t, self.group, "all_reduce" if plain else self.algorithm)

    def exchange_only(self) -> None:
        """The step's collectives alone - same buckets, same order, same dtypes, no compute around them (bench.py `comm`)."""
        if not dist.is_initialized():
            return
        if self.exchange == "bf16":
            work = [self._sum(self.buf[self.n:], plain=True)] + [self._sum(self.buf16[a:b]) for a, b in reversed(self.param_chunks)]
        else:
            work = [self._sum(self.buf[a:b]) for a, b in reversed(self.chunks)]
        for w in work:
            w.wait()

    def bytes_per_step(self) -> int:
        """Payload one rank hands to the collectives per step (the algorithm decides how many times it crosses a link)."""
        return self.n * 2 + (self.buf.numel() - self.n) * 4 if self.exchange == "bf16" else self.buf.numel() * 4

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
        if self.buf16 is not None and self.buf16_filled:
            self.buf16.zero_()

    def all_reduce(self, async_op: bool = False) -> None:
        if not dist.is_initialized():
            return                                   # single process: nothing to exchange
        self._work = [self._sum(self.buf[a:b]) for a, b in self.chunks]
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
            work_tail = self._sum(tail, plain=True)          # (loss, den, num): fp32
            order = list(reversed(self.param_chunks))
            work = []
            for a, b in order:
                if not self.buf16_filled:
                    self.buf16[a:b].copy_(self.buf[a:b])                 # one rounding per rank, on the device
                work.append(self._sum(self.buf16[a:b]))

            def wait(i):
                if i == 0:
                    work_tail.wait()
                work[i].wait()
            optimizer.step_ranges(order, before_each=wait, grads=self.buf16)
            return
        order = list(reversed(self.chunks))
        work = [self._sum(self.buf[a:b]) for a, b in order]
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
            work_tail = self._sum(tail, plain=True)                              # (loss, den, num): fp32
            self.buf16[split:self.n].copy_(self.buf[split:self.n])
            work_first = self._sum(self.buf16[split:self.n]) if self.n > split else None
            run_rest()
            order = [(split, self.n)] + list(reversed(head))
            work = [work_first]
            for a, b in order[1:]:
                self.buf16[a:b].copy_(self.buf[a:b])
                work.append(self._sum(self.buf16[a:b]))

            def wait(i):
                if i == 0:
                    work_tail.wait()
                if work[i] is not None:
                    work[i].wait()
            optimizer.step_ranges(order, before_each=wait, grads=self.buf16)
            return
        work_first = self._sum(self.buf[split:])               # fusion stack + classifier + (loss, den, num)
        run_rest()
        order = [(split, self.buf.numel())] + list(reversed(head))
        work = [work_first] + [self._sum(self.buf[a:b]) for a, b in order[1:]]
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

    def __init__(self, model, optimizer, group=None, n_buckets: int = 4, exchange: str = "fp32", overlap: Optional[bool] = None,
                 algorithm: str = "all_reduce", grad_bf16: Optional[bool] = None):
        """overlap: with more than one rank, run the step in two parts (runtime.Plan.step_part) and put the all-reduce of the
        fusion stack's / classifier's gradients on the wire before the encoders' backward starts (plans that cannot be split -
        fp32 mode - exchange after the whole backward as before).  None = the environment's M2F_DP_OVERLAP (default OFF: the
        path is verified bit for bit against overlap off through gloo staging, tests/test_dp_multirank_gpu.py, but has not yet
        run on RCCL with two devices - no multi-GPU box has been available to this repository).
        algorithm: "all_reduce" | "rs_ag" (GradReducer).
        grad_bf16 (bf16 exchange, whole-step form): the step leaves its gradients rounded in the exchange buffer itself
        (runtime.Plan.grad_bf16) instead of a rounding pass over the fp32 buffer; None = M2F_GRAD_BF16 (default on)."""
        if overlap is None:
            overlap = os.environ.get("M2F_DP_OVERLAP", "0") == "1"
        self.model, self.optimizer, self.overlap = model, optimizer, bool(overlap)
        self.grad_bf16 = (os.environ.get("M2F_GRAD_BF16", "1") != "0") if grad_bf16 is None else bool(grad_bf16)
        self._split: Optional[int] = None
        eng = model.engine()
        eng.ensure_grad()
        self.reducer = GradReducer(eng.flat_grad_ext, eng.flat.numel(), group, n_buckets, exchange, algorithm)
        self.reducer.align_to(o for (_, o, _, _) in eng.items)      # buckets of whole tensors: the optimizer keeps the bf16 parameter shadows current
        optimizer.grad_scale = self.reducer.global_den
        if self.reducer.exchange == "bf16" and self.grad_bf16 and self.reducer.world() > 1 and not self.overlap:
            eng.grad_bf16_buf = self.reducer.buf16           # train plans leave their gradients, rounded once, in the exchange buffer

    def split_for(self, plan) -> int:
        """First element of the flat gradient buffer that is final after part 0 of a step, or 0 when the step is not split.  The
        value is a property of the parameter layout (the fusion stack's first parameter), not of a batch shape, and every rank
        must use the same one or their collectives differ in number and size: it is taken once - from the first plan, or, on a
        rank whose very first shard is empty, from a one-dialogue plan built for the purpose - and a plan that disagrees later
        raises instead of issuing a different schedule."""
        if not (self.overlap and self.reducer.world() > 1):
            return 0
        if self._split is None:
            if plan is None:
                plan = self.model.engine().plan(1, 16, True, self.model.training and self.model.m2f_config.dropout > 0.0)
            self._split = int(plan.split_offset())
        elif plan is not None and int(plan.split_offset()) != self._split:
            raise RuntimeError(f"mer_amd.dp: this plan splits its backward at {plan.split_offset()}, earlier plans at {self._split}; "
                               "the ranks' gradient collectives would no longer match")
        return self._split

    def __call__(self, text, audio, mask, emotion, label_smoothing: float = 0.1, class_weights=None,
                 use_graph: bool = True) -> torch.Tensor:
        eng = self.model.engine()
        B, L = mask.shape
        cur = torch.cuda.current_stream(eng.device)
        eng.stream.wait_stream(cur)
        if B == 0:                                    # empty shard: contribute zeros, but take part in every collective
            with torch.cuda.stream(eng.stream):
                self.reducer.buf16_filled = eng.grad_bf16_buf is not None and eng.grad_bf16_buf is self.reducer.buf16
                self.reducer.zero_contribution()
                eng.publish_grads()
                # the SAME bucket schedule as the ranks that hold dialogues (number, order and sizes of the collectives)
                split = self.split_for(None)
                if split > 0:
                    self.reducer.reduce_and_step_split(self.optimizer, lambda: None, split)
                else:
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
            split = self.split_for(plan)
            # bf16 exchange, whole-step form: the step itself leaves every gradient rounded in the exchange buffer (the weight-gradient
            # launch writes bf16 dW directly: no fp32 dW round trip, no rounding pass over 4 bytes per parameter before the all-reduce) -
            # the engine arms its train plans with the reducer's buffer (model._Engine._arm_grad_bf16; plans that cannot keep fp32)
            self.reducer.buf16_filled = split == 0 and self.reducer.buf16 is not None and getattr(plan, "_g16_ref", None) is self.reducer.buf16
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
