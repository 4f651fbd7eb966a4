"""Data parallelism for the training path: one process per GPU, identical replicas, sum-all-reduce
of gradient buckets per optimiser step (RCCL through ``torch.distributed``; ``gloo`` on CPU for
tests).  The reference trains on a single device (core/trainer.py:143-161); samples are independent
(no batch statistics anywhere on the path), so the gradient all-reduce is the only exchange the path
needs.

All parameter gradients live in ONE flat fp32 buffer (3.65 MB for the default network) and every
``p.grad`` is a view into it: backward writes in place and the collectives need no packing copies.
The buffer is cut into ``buckets`` contiguous ranges of whole parameters.  A backward that knows the
reducer (``attach``: the default network's hand-written backward) writes each weight gradient
straight into its view, tells the reducer (``mark_ready``), and the reducer starts the asynchronous
all-reduce of a bucket as soon as its last gradient has been queued, so the collective runs on the
communication stream beside the rest of the backward pass.  ``all_reduce()`` after backward starts
whatever is still pending, waits, applies the 1/world factor to the buffer once and re-arms the
buckets for the next step (however the caller zeroes the gradients).  Several backward passes per
optimiser step (gradient accumulation) run all but the last under ``no_sync()``; a backward that
starts while a bucket's collective is already in flight is refused, because it would add local
gradients to a partly reduced buffer.
"""
import contextlib
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradAllReducer:
    """``gather=False`` (default): every ``p.grad`` is a view of the flat buffer all the time and backward accumulates
    into it in place.  ``gather=True`` is for networks with hundreds of small parameters (RepRFN's training form has
    264): ``zero_grad`` drops the gradients, backward hands each parameter its gradient tensor without a copy, and
    ``all_reduce`` first moves them all into the buffer with one multi-tensor copy and re-points ``p.grad`` at the
    views, so that the optimiser still reads the buffer.

    ``buckets``: number of contiguous ranges the buffer is reduced in (whole parameters, about equal bytes).
    ``force_collective``: run the collectives even with one rank (tests: proves the RCCL branch on a single GPU)."""

    def __init__(self, params: Iterable[Tensor], process_group=None, gather: bool = False, buckets: int = 1,
                 force_collective: bool = False) -> None:
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradAllReducer: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradAllReducer: parameters must share one device and dtype")
        if buckets < 1:
            raise ValueError("FlatGradAllReducer: buckets must be >= 1")
        self.group = process_group
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.gather = gather
        self.force_collective = force_collective
        self.views: List[Tensor] = []
        self._index: Dict[int, int] = {}
        offs = []
        off = 0
        for i, p in enumerate(self.params):
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = None if gather else self.views[-1]
            self._index[id(p)] = i
            offs.append(off)
            off += p.numel()
        # bucket b = parameters [first_b, first_{b+1}): cut where the running size passes b/buckets of the total
        total = off
        self.bucket_of: List[int] = []
        b = 0
        for i, p in enumerate(self.params):
            if b + 1 < buckets and offs[i] >= (b + 1) * total / buckets:
                b += 1      # at most one step per parameter: no bucket is ever empty
            self.bucket_of.append(b)
        self.n_buckets = max(self.bucket_of) + 1
        self.bucket_range: List[tuple] = []
        for b in range(self.n_buckets):
            idx = [i for i, bb in enumerate(self.bucket_of) if bb == b]
            lo, hi = offs[idx[0]], offs[idx[-1]] + self.params[idx[-1]].numel()
            self.bucket_range.append((lo, hi))
        self._pending = [sum(1 for bb in self.bucket_of if bb == b) for b in range(self.n_buckets)]
        self._left = list(self._pending)
        self._ready = [False] * len(self.params)
        self._works: List[Optional[object]] = [None] * self.n_buckets
        self._launched = [False] * self.n_buckets
        self._defer = False               # inside no_sync(): mark_ready records, nothing is started
        self._log_stale = False           # launch_log belongs to a finished step: cleared by the next launch
        self.launch_log: List[int] = []   # buckets in the order their collective was started (tests, profiling)

    # ------------------------------------------------------------------ bookkeeping
    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _collective_needed(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.world_size > 1 or self.force_collective

    def attach(self, module: torch.nn.Module) -> "FlatGradAllReducer":
        """Let ``module``'s own backward write gradients straight into the views (``grad_view`` / ``mark_ready``)."""
        module._grad_reducer = self
        return self

    def attach_hooks(self) -> "FlatGradAllReducer":
        """Gather mode with overlap, for networks whose backward is assembled from per-operator autograd nodes (the
        temporal network's 113 parameters, RepRFN's 264): a post-accumulate hook on every parameter tells the reducer
        when autograd has finished that parameter's gradient; when a bucket's last one has arrived its gradients move
        into the flat buffer with ONE multi-tensor copy, the parameters' ``.grad`` are re-pointed at the views and the
        bucket's all-reduce starts while autograd is still working through the earlier layers.  Buckets complete in
        reverse parameter order, as backward visits the layers.  Parameters that receive no gradient in a step are
        picked up by ``all_reduce()`` (zeros)."""
        if not self.gather:
            raise ValueError("attach_hooks: the in-place mode is announced by the backward itself (attach / mark_ready)")
        if getattr(self, "_hooks", None):
            return self
        self._hooks = [p.register_post_accumulate_grad_hook(self._grad_arrived) for p in self.params]
        return self

    def _grad_arrived(self, p: Tensor) -> None:
        i = self._index.get(id(p))
        if i is None or self._defer:
            return
        b = self.bucket_of[i]
        if self._launched[b] and self._collective_needed():
            raise RuntimeError("FlatGradAllReducer: a gradient arrived for a bucket that is already being reduced "
                               "(second backward of a step without no_sync()?)")
        if self._ready[i]:
            return
        self._ready[i] = True
        self._left[b] -= 1
        if self._left[b] == 0:
            self._collect_bucket(b)
            self._launch(b)

    def _collect_bucket(self, b: int) -> None:
        src, dst, empty = [], [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if self.bucket_of[i] != b or p.grad is v:
                continue
            if p.grad is None:
                empty.append(v)
            else:
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if empty:
            torch._foreach_zero_(empty)
        if src:
            torch._foreach_copy_(dst, src)

    def grad_view(self, p: Tensor) -> Optional[Tensor]:
        i = self._index.get(id(p))
        return None if i is None else self.views[i]

    def zero_grad(self) -> None:
        """Replaces ``optimizer.zero_grad()``: keeps the views and zeroes the buffer, or (gather mode) drops the
        gradients so that backward assigns instead of accumulating.  Also re-arms the buckets."""
        if self.gather:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()
        self._wait_all()
        self._rearm()
        self.launch_log = []
        self._log_stale = False

    def _rearm(self) -> None:
        self._left = list(self._pending)
        self._ready = [False] * len(self.params)
        self._launched = [False] * self.n_buckets

    def _wait_all(self) -> None:
        for wk in self._works:
            if wk is not None:
                wk.wait()   # RCCL: the current stream waits for the communication stream; gloo: blocks the host
        self._works = [None] * self.n_buckets

    def _in_flight(self) -> bool:
        return any(self._launched) and self._collective_needed()

    def begin_backward(self) -> None:
        """Called by an attached backward before it writes its first gradient.  A pass that starts while a bucket of
        this step is already being reduced would accumulate local gradients into a partly summed buffer and write
        beside a running collective: refused.  Otherwise every parameter counts as not ready again, so that the pass
        that ends the step (the one outside ``no_sync``) announces all of them."""
        if self._in_flight():
            raise RuntimeError("FlatGradAllReducer: a backward pass started after this step's all-reduce had begun; "
                               "run all but the last backward of a step under no_sync(), or call all_reduce() between them")
        self._rearm()

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context add to the buffer and start no collective."""
        prev, self._defer = self._defer, True
        try:
            yield self
        finally:
            self._defer = prev

    def mark_ready(self, p: Tensor) -> None:
        """The gradient of ``p`` for this step is complete in its view (all kernels that write it are queued on the
        current stream).  Starts the bucket's all-reduce when it was the bucket's last one."""
        i = self._index.get(id(p))
        if i is None or self._ready[i]:
            return
        b = self.bucket_of[i]
        if self._launched[b] and self._collective_needed():
            raise RuntimeError("FlatGradAllReducer.mark_ready: the bucket of this parameter is already being reduced "
                               "(second backward of a step without no_sync()?)")
        self._ready[i] = True
        self._left[b] -= 1
        if self._left[b] == 0 and not self._defer:
            self._launch(b)

    def _launch(self, b: int) -> None:
        if self._launched[b]:
            return
        if self._log_stale:
            self.launch_log, self._log_stale = [], False
        self._launched[b] = True
        self.launch_log.append(b)
        if not self._collective_needed():
            return
        lo, hi = self.bucket_range[b]
        chunk = self.flat[lo:hi]
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal of the multi-rank path without RCCL (ranks sharing a GPU): gloo's own device path can stall
            # behind a busy GPU, so the bucket crosses through host memory
            host = chunk.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            chunk.copy_(host)
        else:
            # RCCL: the collective is queued on the process group's communication stream behind everything the current
            # stream holds so far, and runs beside the kernels queued after this call
            self._works[b] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def collect(self) -> None:
        """Gather mode: move the gradients backward produced into the buffer (parameters without one get zeros)."""
        src, dst, empty = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                empty.append(v)
            else:
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if empty:
            torch._foreach_zero_(empty)   # one launch for all of them (RepRFN's training form has dozens)
        if src:
            torch._foreach_copy_(dst, src)

    def check_views(self) -> None:
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("a parameter's .grad no longer aliases the flat bucket "
                                   "(use FlatGradAllReducer.zero_grad, not set_to_none)")

    def all_reduce(self) -> None:
        """Average the gradients over the ranks (call after backward, before optimizer.step): starts the buckets
        backward did not announce, waits for all of them, scales once, re-arms the buckets."""
        if self.gather:
            self.collect()
        for b in reversed(range(self.n_buckets)):   # backward fills the buffer from its end
            self._launch(b)
        needed = self._collective_needed()
        if needed:
            self._wait_all()
            w = self.world_size
            if w > 1:
                self.flat.mul_(1.0 / w)
        # the step's reduction is complete: the next backward starts from armed buckets whether the caller zeroes the
        # gradients through zero_grad(), optimizer.zero_grad(set_to_none=False) or flat.zero_()
        self._rearm()
        self._log_stale = True


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None, force_collective: bool = False) -> None:
    """Make every replica start from rank ``src``'s parameters and buffers (one flat broadcast).  The copy back goes
    through ``Tensor.copy_`` on the parameter itself, which bumps its version counter: packed-weight caches keyed on
    the version see the new values."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(process_group) == 1 and not force_collective:
        return
    tensors = list(module.parameters()) + list(module.buffers())
    if not tensors:
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
        if flat.is_cuda and dist.get_backend(process_group) == "gloo":
            host = flat.cpu()
            dist.broadcast(host, src=src, group=process_group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src, group=process_group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t).to(t.dtype))
            off += t.numel()
    invalidate = getattr(module, "invalidate_packs", None)
    if callable(invalidate):
        invalidate()


def shard_range(total: int, rank: int, world: int) -> range:
    """Indices of the samples / frames rank ``rank`` owns: i, i + world, ... (frame i -> GPU i mod world)."""
    return range(rank, total, world)
