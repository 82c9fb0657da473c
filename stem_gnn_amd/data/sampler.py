"""Mini-batch neighbour sampler with the NeighborLoader contract the reference relies on
(reference pretrain.py:151-153: ``NeighborLoader(data, input_nodes, num_neighbors=[10]*L,
batch_size, shuffle=True)``; SURVEY.md §5 'long-context'): per hop every newly reached node
draws up to ``fanout`` of its in-neighbours uniformly WITHOUT replacement; the subgraph keeps
every sampled edge (neighbour -> node); local numbering puts the seeds first, then nodes in
hop order; ``batch_size`` = number of seeds.

``HipNeighborSampler`` is the fused device sampler (csrc/sampler.hip); a torch-op restatement of the same
contract lives in tests/torch_sampler.py as a test aid.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch
from torch import Tensor

from .. import ops


@dataclass
class Batch:
    batch_size: int
    n_id: Tensor              # int64 [Nb] global node id per local node (seeds first)
    x: Tensor                 # int64 [Nb] rows of node_text_feat
    edge_index: Tensor        # int64 [2, Eb] local ids, row 0 = source, row 1 = target
    xe: Tensor                # int64 [Eb] edge-type ids
    node_text_feat: Tensor
    edge_text_feat: Tensor


class HipNeighborSampler:
    """The same contract on the fused HIP sampler (csrc/sampler.hip): the full graph's by-target
    CSR and edge types stay resident in HBM, a batch is a dozen small launches + one 8-byte
    device->host copy, and the batch arrives with its by-target CSR already built
    (``Batch.graph``)."""

    def __init__(self, edge_index: Tensor, xe: Tensor, num_nodes: int, x: Tensor, node_text_feat: Tensor,
                 edge_text_feat: Tensor, num_neighbors: List[int], seed: int = 0):
        self.num_nodes = num_nodes
        self.fanouts = [int(f) for f in num_neighbors]
        rowptr, src, eid, _ = ops.csr_build(edge_index.contiguous(), num_nodes, 1)
        self.rowptr, self.src = rowptr, src
        self.etype = ops.gather_i32(xe.to(torch.int32).contiguous(), eid)
        self.x, self.ntf, self.etf = x, node_text_feat, edge_text_feat
        self.local_of = ops.sampler_init_map(num_nodes, edge_index.device)
        self.seed, self.calls = int(seed), 0
        # degree bounds of every batch, known on the host once (one sync here, none per batch): a
        # batch node is expanded at most once, and keeps at most its full-graph out-edges
        if num_nodes and edge_index.size(1):
            d_in, d_out = torch.stack([(rowptr[1:] - rowptr[:-1]).max().long(),
                                       torch.bincount(edge_index[0], minlength=1).max()]).tolist()
        else:
            d_in = d_out = 0
        cap = max((f if f >= 0 else d_in) for f in self.fanouts) if self.fanouts else 0
        self.batch_max_in_degree, self.batch_max_out_degree = int(min(cap, d_in)), int(d_out)

    def sample(self, seeds: Tensor) -> Batch:
        from ..graph import GraphStructure
        self.calls += 1
        n_id, rowptr, src, etype, coo, nb, eb, ab = ops.sample_batch(self.rowptr, self.src, self.etype, self.num_nodes,
                                                                 seeds.contiguous(), self.fanouts, self.seed,
                                                                 self.calls * 64, self.local_of)
        n_id64 = n_id.long()
        b = Batch(batch_size=seeds.numel(), n_id=n_id64, x=self.x.index_select(0, n_id64), edge_index=coo, xe=etype.long(),
                  node_text_feat=self.ntf, edge_text_feat=self.etf)
        b.graph = GraphStructure.from_csr(rowptr, src, coo, nb, etype_slot=etype,
                                          max_in_degree=self.batch_max_in_degree,
                                          max_out_degree=self.batch_max_out_degree, active_rows=ab)
        return b


class NeighborLoader:
    """Iterates shuffled seed batches (one epoch), sharded round-robin across ranks."""

    def __init__(self, sampler, input_nodes: Tensor, batch_size: int, shuffle: bool = True,
                 rank: int = 0, world_size: int = 1, seed: int = 0):
        self.sampler, self.batch_size = sampler, batch_size
        nodes = input_nodes
        if shuffle:
            g = torch.Generator(device=nodes.device).manual_seed(seed)  # shared seed: same order on all ranks
            nodes = nodes[torch.randperm(nodes.numel(), generator=g, device=nodes.device)]
        self.nodes = nodes[rank::world_size]

    def __len__(self):
        return (self.nodes.numel() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, self.nodes.numel(), self.batch_size):
            yield self.sampler.sample(self.nodes[i:i + self.batch_size])


class MixLoader:
    """The loader of a multi-dataset mix (reference pretrain.py:144-153): the weighted seed list is REBUILT at the start
    of every epoch (``get_train_node_idx``: random shares of the members with fractional weights change per epoch),
    shuffled with a seed shared by all ranks, dealt round-robin to the ranks, and sampled in batches.  ``ptr`` /
    ``weights``: member node offsets and seed weights (data/multi.py)."""

    def __init__(self, sampler, ptr: Tensor, weights, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0,
                 device=None):
        from .multi import get_train_node_idx
        self._build = get_train_node_idx
        self.sampler, self.ptr, self.weights, self.batch_size = sampler, ptr, list(weights), batch_size
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        self.device = device if device is not None else sampler.rowptr.device
        self.num_seeds = None

    def _epoch_nodes(self) -> Tensor:
        g = torch.Generator(device=self.device).manual_seed(self.seed + 7919 * self.epoch)  # same list on every rank
        nodes = self._build(self.ptr, self.weights, device=self.device, generator=g)
        nodes = nodes[torch.randperm(nodes.numel(), generator=g, device=self.device)]
        self.num_seeds = int(nodes.numel())
        return nodes[self.rank::self.world_size]

    def __iter__(self):
        nodes = self._epoch_nodes()
        self.epoch += 1
        for i in range(0, nodes.numel(), self.batch_size):
            yield self.sampler.sample(nodes[i:i + self.batch_size])


class PrefetchLoader:
    """Wraps a loader of device-resident batches (e.g. NeighborLoader over a HipNeighborSampler): batch i+1 is
    sampled on a side HIP stream BEFORE the caller enqueues step i, so the sampler's launches and its one
    device->host size read overlap the steps already queued on the current stream instead of holding them up.
    ``prepare(batch)`` (optional) runs on the side stream too, e.g. the feature gather and the transposed CSR.

    ``uses`` names the batch attributes the consumer's kernels read (tensors or objects with ``record_stream``, like
    GraphStructure): they were allocated on the side stream, so they are handed to the current stream with
    ``record_stream`` -- only those, because every recorded block turns its later free into an event round trip."""

    def __init__(self, loader, device, prepare=None, uses=("feat", "x", "xe", "graph")):
        self.loader, self.device, self.prepare, self.uses = loader, torch.device(device), prepare, tuple(uses)
        self.side = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _load(self, it):
        with torch.cuda.stream(self.side):
            b = next(it, None)
            if b is not None and self.prepare is not None:
                b = self.prepare(b) or b
            return b

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._load(it)
        while nxt is not None:
            cur = nxt
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self.side)
            for name in self.uses:
                v = getattr(cur, name, None)
                if v is not None and (not isinstance(v, Tensor) or v.is_cuda):
                    v.record_stream(main)
            nxt = self._load(it)  # before the caller's step: it overlaps the steps still queued on the device
            yield cur
