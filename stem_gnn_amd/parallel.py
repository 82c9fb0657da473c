"""Data-parallel plumbing for the pretraining path (no reference counterpart: the reference is
single-process, pretrain.py:84).  One process per GPU; RCCL over xGMI through
torch.distributed's "nccl" backend on ROCm; "gloo" on CPU for the logic tests.

The path shards naturally over seed-node mini-batches: graph structure and
feature tables are replicated, every rank samples its own subgraphs from its shard of the
shuffled seed list and runs the full step; the only exchange is the gradient all-reduce
(DistributedDataParallel buckets, overlapped with backward) plus, when EMA codebook updates
are enabled, the VQ statistics all-reduce the reference already carries (vq.py:666,672).
"""
from __future__ import annotations

import os
from typing import Iterable, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def init_distributed(backend: str = "nccl", device=None) -> Tuple[int, int]:
    """(rank, world_size) from the torchrun environment; no-op when WORLD_SIZE <= 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kwargs)
    return rank, world


def shard_seeds(nodes: Tensor, rank: int, world_size: int, seed: int, shuffle: bool = True) -> Tensor:
    """Shuffle the seed list with a seed shared by all ranks, then deal it round-robin: the
    shards are disjoint and cover the list; shard sizes differ by at most one."""
    if shuffle:
        g = torch.Generator(device=nodes.device).manual_seed(seed)
        nodes = nodes[torch.randperm(nodes.numel(), generator=g, device=nodes.device)]
    return nodes[rank::world_size]


def reduce_bench_stats(elapsed_s: float, units: float, device) -> Tuple[float, float]:
    """(max over ranks of the elapsed time, sum over ranks of the processed units)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t), float(u)


def allreduce_mean_grads_(params: Iterable[Tensor]) -> None:
    """Explicit (non-overlapped) alternative to DDP: average all gradients in ONE flat
    all-reduce (2.2 MB at D=128, 41.7 MB at D=768: SURVEY.md §8e)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class FlatGradSync:
    """The explicit form of the gradient exchange for ``pretrain_step(..., grad_sync=...)``: after backward the trainable
    gradients are packed into ONE flat buffer by a fused multi-tensor copy, averaged by ONE all-reduce (RCCL ``AVG``; sum +
    scale on backends without it), and every ``p.grad`` becomes a VIEW of the flat buffer -- clipping and FusedAdamW read
    the views, nothing is copied back.  Two launches plus the collective per step, where DistributedDataParallel's
    reducer spends 23 (one copy into its bucket view per parameter, every step, because the step hands autograd fresh
    gradient tensors: +0.11 ms on the 1.5 ms C4 step at world size 1, bench.py ``dp1_under_ddp``).  The price: the
    reduction starts when the backward has ended instead of overlapping its tail; for this model's 2.2 MB (D = 128) that
    is one short collective.  The EMA teacher's parameters never receive gradients and are not exchanged; parameters
    that received none in a step (``no_codebook=True``) contribute zeros."""

    def __init__(self, params: Iterable[Tensor]):
        self.params = [p for p in params if p.requires_grad]
        self.flat = None
        self.views = None

    def _buffers(self):
        if self.flat is None:
            p0 = self.params[0]
            total = sum(p.numel() for p in self.params)
            self.flat = torch.empty(total, dtype=p0.dtype, device=p0.device)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self.flat, self.views

    def __call__(self) -> None:
        if not self.params:
            return
        flat, views = self._buffers()
        src, dst = [], []
        for p, v in zip(self.params, views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        if dist.is_initialized():  # (also at world size 1: bench.py's readiness proxy wants the collective's launch cost)
            if dist.get_backend() == "nccl":
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.div_(dist.get_world_size())
        for p, v in zip(self.params, views):
            p.grad = v


def ddp_bucket_cap_mb(model: torch.nn.Module, buckets: int = 3) -> float:
    """Bucket size that cuts the trainable gradient into about ``buckets`` reductions.  The step's backward produces
    its gradients in three stretches -- the decoders / heads, the quantiser, the encoder (reverse registration order,
    which is how the reducer fills buckets) -- so with three buckets the heads' and the quantiser's gradients are on
    the wire while the encoder's backward still runs.  torch's default (25 MB) puts the whole 2.2 MB gradient of the
    D = 128 model, and most of the 41.7 MB of the D = 768 one, into a single reduction behind the last weight gradient."""
    total = sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad)
    return min(25.0, max(total / max(int(buckets), 1) / 2 ** 20, 1.0 / 64))


def wrap_ddp(model: torch.nn.Module, device_index=None, find_unused_parameters=False, bucket_cap_mb=None):
    """DistributedDataParallel over the trainable parameters.  The EMA teacher never receives
    gradients (reference pt_model.py:93 detaches it) and is excluded; BatchNorm statistics stay
    per-rank (no buffer broadcast), like the per-batch statistics of the single-GPU path.

    ``find_unused_parameters``: pass True for runs in which some trainable parameters receive no gradient in a step
    (``forward(..., no_codebook=True)`` bypasses the VQ projections; a zero loss weight drops a decoder): the reducer
    then searches the graph for them instead of raising.  The default path (every parameter used, MoE routing soft)
    runs without the search.

    ``bucket_cap_mb``: None = ``ddp_bucket_cap_mb(model)`` (three reductions per step, overlapped with backward; the
    reducer re-cuts its buckets after the first step, which still runs as one)."""
    if hasattr(model, "sem_encoder"):
        for p in model.sem_encoder.parameters():
            p.requires_grad_(False)
    cap = ddp_bucket_cap_mb(model) if bucket_cap_mb is None else float(bucket_cap_mb)
    kw = dict(broadcast_buffers=False, gradient_as_bucket_view=True, find_unused_parameters=bool(find_unused_parameters),
              bucket_cap_mb=cap)
    if device_index is not None:
        kw["device_ids"] = [device_index]
    ddp = torch.nn.parallel.DistributedDataParallel(model, **kw)
    # Without a communication hook the reducer divides EVERY gradient by the world size with a launch of its own as it
    # becomes ready: 23 five-microsecond launches per step on this model (rocprofv3, world size 1: +0.12 ms on a 1.5 ms
    # step -- 8 % of scaling efficiency before a byte has moved).  The reducer's BUILT-IN all-reduce hook (C++; a Python
    # hook costs more than it saves: measured +0.2 ms) divides once per BUCKET -- three launches per step -- with the same
    # arithmetic: pre-divide, then sum.
    ddp._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
    return ddp
