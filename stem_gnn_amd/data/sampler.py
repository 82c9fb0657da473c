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
from typing import List, Optional

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
    cap_nodes: int = 0        # the most nodes a batch of this many seeds / these fan-outs can have (buffer sizing)


MAX_FANOUT = 32  # csrc/sampler.hip kMaxFanout: a row's picks live in a register array of this size


def _check_fanouts(num_neighbors) -> List[int]:
    """-1 (every in-neighbour) or 1 .. MAX_FANOUT per hop; anything else is rejected here, before a launch."""
    f = [int(v) for v in num_neighbors]
    bad = [v for v in f if v == 0 or v < -1 or v > MAX_FANOUT]
    if bad:
        raise ValueError(f"fan-out must be -1 or 1..{MAX_FANOUT} per hop, got {f}")
    return f


class HipNeighborSampler:
    """The same contract on the fused HIP sampler (csrc/sampler.hip): the full graph's by-target
    CSR and edge types stay resident in HBM, a batch is 13 launches (two hops) whose sizes (12 bytes) land in pinned
    host memory, and the batch arrives with BOTH CSR views, 1 / in-degree and its int64 id / type / feature-row
    vectors already built (``Batch.graph``): no launch is left for the consumer but the feature gather."""

    def __init__(self, edge_index: Tensor, xe: Tensor, num_nodes: int, x: Tensor, node_text_feat: Tensor,
                 edge_text_feat: Tensor, num_neighbors: List[int], seed: int = 0):
        self.num_nodes = num_nodes
        self.fanouts = _check_fanouts(num_neighbors)
        rowptr, src, eid, _ = ops.csr_build(edge_index.contiguous(), num_nodes, 1)
        # a graph without edges still gets a one-slot array: the library takes a null source array for a bad argument
        self.rowptr, self.src = rowptr, (src if src.numel() else torch.zeros(1, dtype=torch.int32, device=src.device))
        self.etype = ops.gather_i32(xe.to(torch.int32).contiguous(), eid) if eid.numel() else None
        self.x, self.ntf, self.etf = x.contiguous(), node_text_feat, edge_text_feat
        self._slots = None  # identity slot -> edge map, shared by every batch (slices of one arange)
        self._sizes = None  # pinned landing zone of the batches' sizes
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

    def with_fanouts(self, num_neighbors: List[int]) -> "HipNeighborSampler":
        """A sampler over the SAME resident graph (CSR and edge types shared) with other fan-outs -- the reference
        builds its training loader ([10] * L or [30] * L) and its evaluation loader ([-1] * L) over one ``data`` object
        (utils/loader.py:10-25).  The derived sampler gets its OWN scratch map and draw counter: a training loader's
        prefetched batch may still be in flight on its side stream when the evaluation sampler starts claiming entries
        on the main stream, and nothing orders the two."""
        other = object.__new__(HipNeighborSampler)
        other.__dict__.update(self.__dict__)
        other.fanouts = _check_fanouts(num_neighbors)
        other.local_of = ops.sampler_init_map(self.num_nodes, self.rowptr.device)
        other.seed, other.calls = self.seed + 0x9E3779B1, 0
        d_in = int((self.rowptr[1:] - self.rowptr[:-1]).max()) if self.num_nodes else 0
        cap = max((f if f >= 0 else d_in) for f in other.fanouts) if other.fanouts else 0
        other.batch_max_in_degree = int(min(cap, d_in))
        other._slots = other._sizes = None
        return other

    def sample_async(self, seeds: Tensor) -> "PendingBatch":
        """Enqueue the batch's launches on the current stream and return at once; ``result()`` waits for its sizes
        (12 bytes) only.  A loader that keeps one batch in flight beyond the one being prepared never waits.
        (Fan-outs with a -1 take the sized-per-hop path, which reads one size per hop: ``result()`` is immediate.)"""
        self.calls += 1
        if any(f < 0 for f in self.fanouts):
            o = ops.sample_batch_full(self.rowptr, self.src, self.etype, self.num_nodes, seeds.contiguous(), self.fanouts,
                                      self.seed, self.calls * 64, self.local_of, self.x)
            return PendingBatch(self, _Ready(o), seeds.numel())
        if self._sizes is None:
            self._sizes = torch.empty(8, 3, dtype=torch.int32, pin_memory=True)  # ring: at most a few batches in flight
        # the by-source view's ordering pass walks a row in one thread: only for graphs whose out-rows are short
        pend = ops.sample_batch_views_launch(self.rowptr, self.src, self.etype, self.num_nodes, seeds.contiguous(),
                                             self.fanouts, self.seed, self.calls * 64, self.local_of, self.x,
                                             counts_host=self._sizes[self.calls % 8],
                                             by_source=self.batch_max_out_degree <= 128)
        return PendingBatch(self, pend, seeds.numel())

    def _adopt(self, o: dict, batch_size: int) -> Batch:
        from ..graph import GraphStructure
        eb = o["eb"]
        if self._slots is None or self._slots.numel() < eb:
            cap, level = 0, batch_size
            for f in self.fanouts:
                level *= max(f, 0)  # a -1 hop has no bound: the identity map just grows with the largest batch seen
                cap += level
            self._slots = torch.arange(max(cap, eb), dtype=torch.int32, device=o["src"].device)
        b = Batch(batch_size=batch_size, n_id=o["n_id64"], x=o["x"], edge_index=o["coo"], xe=o["type64"],
                  node_text_feat=self.ntf, edge_text_feat=self.etf, cap_nodes=o["cap_nodes"])
        b.graph = GraphStructure.from_csr(o["rowptr"], o["src"], o["coo"], o["nb"], etype_slot=o["type"],
                                          max_in_degree=self.batch_max_in_degree,
                                          max_out_degree=self.batch_max_out_degree, active_rows=o["ab"],
                                          eid=self._slots[:eb],
                                          by_source=None if o["rowptr_t"] is None else
                                          (o["rowptr_t"], o["dst_t"], o["eid_t"], o["type_t"], o["inv_deg"]))
        return b

    def sample(self, seeds: Tensor) -> Batch:
        return self.sample_async(seeds).result()


class _Ready:
    def __init__(self, o: dict):
        self.o = o

    def result(self) -> dict:
        return self.o


class PendingBatch:
    """A batch whose launches are enqueued; ``result()`` -> Batch.  ``post`` (optional): what the loader still attaches
    to the finished batch (labels)."""

    def __init__(self, sampler: HipNeighborSampler, pend, batch_size: int):
        self.sampler, self.pend, self.batch_size, self.post = sampler, pend, batch_size, None

    def result(self) -> Batch:
        b = self.sampler._adopt(self.pend.result(), self.batch_size)
        if self.post is not None:
            self.post(b)
        return b


class NeighborLoader:
    """Iterates shuffled seed batches (one epoch), sharded round-robin across ranks.  ``y`` (optional, one row per node
    of the full graph): every batch carries ``batch.y = y[batch.n_id]``, what PyG's loader does with a node-level
    attribute of ``data`` (the finetune loops read ``batch.y[:batch_size]``, reference task/node.py:28,77)."""

    def __init__(self, sampler, input_nodes: Tensor, batch_size: int, shuffle: bool = True,
                 rank: int = 0, world_size: int = 1, seed: int = 0, y: Optional[Tensor] = None):
        self.sampler, self.batch_size, self.y = sampler, batch_size, y
        nodes = input_nodes
        if shuffle:
            g = torch.Generator(device=nodes.device).manual_seed(seed)  # shared seed: same order on all ranks
            nodes = nodes[torch.randperm(nodes.numel(), generator=g, device=nodes.device)]
        self.nodes = nodes[rank::world_size]

    def __len__(self):
        return (self.nodes.numel() + self.batch_size - 1) // self.batch_size

    def _attach(self, b: Batch) -> None:
        if self.y is not None:
            b.y = self.y.to(b.n_id.device)[b.n_id]

    def __iter__(self):
        for p in self.iter_pending():
            yield p.result()

    def iter_pending(self):
        """The same batches as handles whose launches are enqueued (``result()`` -> Batch): see PrefetchLoader."""
        for i in range(0, self.nodes.numel(), self.batch_size):
            p = self.sampler.sample_async(self.nodes[i:i + self.batch_size])
            p.post = self._attach
            yield p


class LinkNeighborLoader:
    """LinkNeighborLoader as the reference uses it (utils/loader.py:28-48; task/link.py:55-61 reads ``edge_label_index``
    and ``edge_label``): batches of LABELLED EDGES; the seeds of a batch are the distinct endpoints of its edges, their
    neighbourhoods are sampled like a node batch's, and ``batch.edge_label_index`` addresses the batch's local node ids
    (``batch.input_id`` = positions of the batch's edges in ``edge_label_index``)."""

    def __init__(self, sampler, edge_label_index: Tensor, edge_label: Tensor, batch_size: int, shuffle: bool = True,
                 seed: int = 0):
        self.sampler, self.batch_size = sampler, batch_size
        self.edge_label_index, self.edge_label = edge_label_index, edge_label
        ids = torch.arange(edge_label_index.size(1), device=edge_label_index.device)
        if shuffle:
            g = torch.Generator(device=ids.device).manual_seed(seed)
            ids = ids[torch.randperm(ids.numel(), generator=g, device=ids.device)]
        self.ids = ids

    def __len__(self):
        return (self.ids.numel() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, self.ids.numel(), self.batch_size):
            ids = self.ids[i:i + self.batch_size]
            pairs = self.edge_label_index[:, ids]
            seeds, inverse = torch.unique(pairs.reshape(-1), return_inverse=True)
            b = self.sampler.sample(seeds)
            b.edge_label_index = inverse.view(2, -1)
            b.edge_label = self.edge_label.to(ids.device)[ids]
            b.input_id = ids
            yield b


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

    def iter_pending(self):
        nodes = self._epoch_nodes()
        self.epoch += 1
        for i in range(0, nodes.numel(), self.batch_size):
            yield self.sampler.sample_async(nodes[i:i + self.batch_size])


class PrefetchLoader:
    """Wraps a loader of device-resident batches (e.g. NeighborLoader over a HipNeighborSampler) and runs it on a side
    HIP stream ahead of the consumer.  With a loader that offers ``iter_pending()`` the pipeline is two deep: while the
    caller enqueues step i, batch i+1 is being prepared (its sizes, read on the host, arrived a step ago -- no wait) and
    the sampler launches of batch i+2 are enqueued, so the host never blocks on the device and the sampler's kernels
    overlap the steps already queued.  Without it, batch i+1 is sampled (one blocking size read) before step i is
    enqueued.  ``prepare(batch)`` (optional) runs on the side stream too, e.g. the feature gather.

    ``uses`` names the batch attributes the consumer's kernels read (tensors or objects with ``record_stream``, like
    GraphStructure): they were allocated on the side stream, so they are handed to the current stream with
    ``record_stream`` -- only those, because every recorded block turns its later free into an event round trip."""

    def __init__(self, loader, device, prepare=None, uses=("feat", "x", "xe", "graph")):
        self.loader, self.device, self.prepare, self.uses = loader, torch.device(device), prepare, tuple(uses)
        self.side = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _launch(self, it):
        with torch.cuda.stream(self.side):
            return next(it, None)

    def _finish(self, pending):
        if pending is None:
            return None
        with torch.cuda.stream(self.side):
            b = pending.result() if isinstance(pending, PendingBatch) else pending
            if self.prepare is not None:
                b = self.prepare(b) or b
            b.ready = torch.cuda.Event()
            b.ready.record(self.side)
            return b

    def __iter__(self):
        pending_iter = getattr(self.loader, "iter_pending", None)
        it = pending_iter() if pending_iter is not None else iter(self.loader)
        nxt = self._finish(self._launch(it))
        ahead = self._launch(it)  # launches in flight, sizes not read yet
        try:
            while nxt is not None:
                cur = nxt
                main = torch.cuda.current_stream(self.device)
                main.wait_event(cur.ready)
                seen = set()  # one record per allocation: a sampler batch is views of one slab (+ the prepared features)
                for name in self.uses:
                    v = getattr(cur, name, None)
                    if isinstance(v, Tensor):
                        if v.is_cuda and v.untyped_storage().data_ptr() not in seen:
                            seen.add(v.untyped_storage().data_ptr())
                            v.record_stream(main)
                    elif v is not None:
                        v.record_stream(main, seen)
                # before the caller's step, so that all of it overlaps the steps still queued on the device
                nxt = self._finish(ahead)
                ahead = self._launch(it) if nxt is not None else None
                yield cur
        finally:
            # a consumer that stops early (break, exception, generator closed) leaves batches in flight on the side
            # stream; whoever uses the sampler next -- on any stream -- must come after them: the sampler's scratch map is
            # only back in its idle state when a batch's last launch has run
            torch.cuda.current_stream(self.device).wait_stream(self.side)
