"""Data parallelism for the training path: one process per GPU, identical replicas, one
sum-all-reduce of the gradients per optimiser step (RCCL through ``torch.distributed``; ``gloo`` on
CPU for tests).  The reference trains on a single device (core/trainer.py:143-161); samples are
independent (no batch statistics anywhere on the path), so the gradient all-reduce is the only
exchange the path needs.

All parameter gradients live in ONE flat fp32 buffer (3.65 MB for the default network) and every
``p.grad`` is a view into it: backward writes in place, the all-reduce is a single collective with
no packing copies, and the 1/world factor is applied to the flat buffer once.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradAllReducer:
    """``gather=False`` (default): every ``p.grad`` is a view of the bucket all the time and backward accumulates into
    it in place (one small add kernel per parameter).  ``gather=True`` is for networks with hundreds of small
    parameters (RepRFN's training form has 264): ``zero_grad`` drops the gradients, backward hands each parameter its
    gradient tensor without a copy, and ``all_reduce`` first moves them all into the bucket with one multi-tensor copy
    and re-points ``p.grad`` at the views, so that the optimiser still reads the bucket."""

    def __init__(self, params: Iterable[Tensor], process_group=None, gather: bool = False) -> None:
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradAllReducer: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradAllReducer: parameters must share one device and dtype")
        self.group = process_group
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.gather = gather
        self.views: List[Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = None if gather else self.views[-1]
            off += p.numel()

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self) -> None:
        """Replaces ``optimizer.zero_grad()``: keeps the views and zeroes the bucket, or (gather mode) drops the
        gradients so that backward assigns instead of accumulating."""
        if self.gather:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()

    def collect(self) -> None:
        """Gather mode: move the gradients backward produced into the bucket (parameters without one get zeros)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                v.zero_()
            else:
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def check_views(self) -> None:
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("a parameter's .grad no longer aliases the flat bucket "
                                   "(use FlatGradAllReducer.zero_grad, not set_to_none)")

    def all_reduce(self) -> None:
        """Average the gradients over the ranks (call after backward, before optimizer.step)."""
        if self.gather:
            self.collect()
        w = self.world_size
        if w == 1:
            return
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal of the multi-rank path without RCCL (ranks sharing a GPU): gloo's own device path can stall
            # behind a busy GPU, so the bucket crosses through host memory
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / w)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Make every replica start from rank ``src``'s parameters and buffers (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
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


def shard_range(total: int, rank: int, world: int) -> range:
    """Indices of the samples / frames rank ``rank`` owns: i, i + world, ... (frame i -> GPU i mod world)."""
    return range(rank, total, world)
