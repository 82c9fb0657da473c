"""Graph-structure object: the int32 CSR views of one (sub)graph that the HIP aggregation
kernels consume, built once per graph on the device.

The reference has no such object: PyG's MessagePassing.propagate re-indexes the raw int64
``edge_index`` on every layer call (reference model/encoder.py:82).  ``GraphStructure`` may be
passed anywhere the modules accept ``edge_index``; a raw tensor is converted through a small
identity-keyed cache so a caller that only knows the reference API pays the build once per
distinct ``edge_index`` tensor.  A loader can hand batches over as GraphStructure directly.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from typing import Optional

import torch
from torch import Tensor

from . import ops

_VALIDATE = True


def set_validation(flag: bool) -> None:
    """Range-check edge_index on every build (one device->host sync per build).  Invalid
    edges can never make a kernel read out of bounds (they are dropped from the CSR); with
    validation on they raise IndexError like an out-of-range index would in the reference."""
    global _VALIDATE
    _VALIDATE = bool(flag)
    ops.set_gather_validation(flag)  # the feature-row lookups follow the same policy


class EdgeTypeAttr:
    """Edge attributes given as (type table [T, D], type id per edge [E]) instead of the dense
    [E, D] tensor ``edge_text_feat[xe]`` the reference materialises on the host
    (reference pretrain.py:38).  ``dense()`` yields exactly that tensor."""

    def __init__(self, table: Tensor, etype: Optional[Tensor]):
        if etype is not None and etype.dtype not in (torch.int32, torch.int64):
            raise RuntimeError("EdgeTypeAttr: etype must be an integer tensor")
        self.table = table.contiguous()
        self.etype = None if etype is None else etype.contiguous()

    def size(self, dim: int) -> int:
        return (self.etype.numel(), self.table.size(1))[dim]

    def __getitem__(self, idx) -> "EdgeTypeAttr":
        return EdgeTypeAttr(self.table, self.etype[idx])

    def dense(self) -> Tensor:
        return self.table.index_select(0, self.etype.long())

    def to(self, device) -> "EdgeTypeAttr":
        return EdgeTypeAttr(self.table.to(device), None if self.etype is None else self.etype.to(device))


class GraphStructure:
    """CSR grouped by target (forward aggregation) and, lazily, by source (its backward).

    ``num_edges`` is the number of rows a dense ``edge_attr`` for this graph has (slots address
    it through ``eid``); for an augmented graph made by ``dropout_undirected`` that is the
    ORIGINAL graph's edge count and the live edge count stays on the device (rowptr[-1])."""

    def __init__(self, edge_index: Optional[Tensor], num_nodes: int, edge_type: Optional[Tensor] = None,
                 validate: Optional[bool] = None):
        self.num_nodes = int(num_nodes)
        self.rowptr_t = self.dst_t = self.eid_t = self.inv_deg = None
        self.etype_slot = self.etype_slot_t = None
        self._edge_type = None
        self._edge_index = None
        # Host-side upper bounds of the largest in/out degree (None = unknown).  Known-small bounds
        # let the aggregation skip the heavy-row split passes (ops.SplitPlan) without a device sync.
        self.max_in_degree: Optional[int] = None
        self.max_out_degree: Optional[int] = None
        # Rows >= active_rows are promised to have no in-edges (a neighbour-sampled batch: only the nodes that were
        # expanded, which come first, receive edges).  The aggregation then computes -- and its consumers read --
        # rows [0, active_rows) only.  None = no promise, every row is computed.
        self.active_rows: Optional[int] = None
        self._plan_in = self._plan_out = None
        # CSR slots the structure may hold (>= its live edge count): num_edges, except for an augmented graph, whose
        # arrays hold up to two slots (edge + mirror) per original edge.  Sizes the heavy-row split plan.
        self.slot_capacity = 0
        if edge_index is None:  # filled in by a factory (dropout_undirected)
            self.num_edges = 0
            self.rowptr = self.src = self.eid = self._bad = None
            return
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError(f"edge_index: expected shape [2, E], got {tuple(edge_index.shape)}")
        self._edge_index = edge_index.contiguous()
        self.num_edges = self.slot_capacity = int(edge_index.size(1))
        self.rowptr, self.src, self.eid, self._bad = ops.csr_build(self._edge_index, self.num_nodes, 1)
        if edge_type is not None:
            self.set_edge_type(edge_type)
        if _VALIDATE if validate is None else validate:
            # one sync: the bad-entry count and, while at it, the degree bounds
            deg_in = (self.rowptr[1:] - self.rowptr[:-1]).max().reshape(1) if self.num_nodes else self._bad.reshape(1) * 0
            deg_out = (torch.bincount(self._edge_index[0].clamp(0, max(self.num_nodes - 1, 0)),
                                      minlength=1).max().reshape(1) if self.num_edges else deg_in * 0)
            bad, d_in, d_out = torch.cat([self._bad.reshape(1).long(), deg_in.long(), deg_out.long()]).tolist()
            if bad:
                raise IndexError(f"edge_index has {bad} entries outside [0, {self.num_nodes})")
            self.max_in_degree, self.max_out_degree = int(d_in), int(d_out)

    @classmethod
    def from_csr(cls, rowptr: Tensor, src: Tensor, edge_index: Tensor, num_nodes: int,
                 etype_slot: Optional[Tensor] = None, max_in_degree: Optional[int] = None,
                 max_out_degree: Optional[int] = None, active_rows: Optional[int] = None,
                 validate: bool = False, eid: Optional[Tensor] = None, by_source=None) -> "GraphStructure":
        """Adopt a by-target CSR whose slot j IS edge j of `edge_index` (what the HIP sampler emits).  ``eid``: the
        identity slot -> edge map when the caller keeps one (else one is made); ``by_source`` = (rowptr_t, dst_t, eid_t,
        etype_slot_t, inv_deg): the transposed view when the caller built it already (ensure_transpose is then a
        no-op)."""
        g = cls(None, num_nodes)
        g.max_in_degree, g.max_out_degree = max_in_degree, max_out_degree
        if active_rows is not None and 0 <= active_rows < num_nodes:
            if validate and int(rowptr[active_rows].item()) != int(rowptr[-1].item()):
                raise RuntimeError("from_csr: rows >= active_rows must have no in-edges")
            g.active_rows = int(active_rows)
        g.num_edges = g.slot_capacity = int(src.numel())
        g._edge_index = edge_index
        g.rowptr, g.src = rowptr, src
        g.eid = torch.arange(g.num_edges, dtype=torch.int32, device=src.device) if eid is None else eid
        g._bad = None
        if etype_slot is not None:
            g._edge_type = etype_slot  # edge j == slot j
            g.etype_slot = etype_slot
        if by_source is not None:
            g.rowptr_t, g.dst_t, g.eid_t, g.etype_slot_t, g.inv_deg = by_source
        return g

    @property
    def edge_index(self) -> Tensor:
        """int64 COO [2, E] (row 0 = source, row 1 = target).  For an augmented graph it is
        rebuilt from the CSR on demand (one device->host sync for the size)."""
        if self._edge_index is None:
            n_live = int(self.rowptr[-1].item())
            deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
            dst = torch.repeat_interleave(torch.arange(self.num_nodes, device=deg.device), deg)
            self._edge_index = torch.stack([self.src[:n_live].long(), dst], dim=0)
        return self._edge_index

    def live_edges_host(self) -> int:
        """Number of edges the CSR actually holds, on the host (one sync the first time for an
        augmented graph; measurement/reporting use only)."""
        if getattr(self, "_live", None) is None:
            self._live = self.num_edges if self._edge_index is not None else int(self.rowptr[-1].item())
        return self._live

    def has_edge_type(self) -> bool:
        return self.etype_slot is not None

    def set_edge_type(self, edge_type: Tensor) -> None:
        if edge_type.numel() != self.num_edges:
            raise RuntimeError("edge_type: one id per edge expected")
        self._edge_type = edge_type.to(torch.int32).contiguous()
        self.etype_slot = ops.gather_i32(self._edge_type, self.eid) if self.num_edges else self._edge_type
        self.etype_slot_t = None
        if self.eid_t is not None:
            self.etype_slot_t = ops.gather_i32(self._edge_type, self.eid_t) if self.num_edges else self._edge_type

    def ensure_transpose(self) -> "GraphStructure":
        if self.rowptr_t is None:
            self.rowptr_t, self.dst_t, self.eid_t, _ = ops.csr_build(self._edge_index, self.num_nodes, 0)
            self.inv_deg = ops.inv_degree(self.rowptr)
        if self._edge_type is not None and self.etype_slot_t is None:
            self.etype_slot_t = (ops.gather_i32(self._edge_type, self.eid_t) if self.num_edges else self._edge_type)
        return self

    def split_plan(self, side: str):
        """The heavy-row split plan for the by-target ("in") or by-source ("out") CSR, or None when
        the host-side degree bound says no row is heavy."""
        bound = self.max_in_degree if side == "in" else self.max_out_degree
        if self.num_edges == 0 or (bound is not None and bound <= max(ops.SPLIT_HEAVY, ops.SPLIT_CHUNK)):
            return None
        if side == "out" and self.rowptr_t is self.rowptr:  # symmetric graph: one degree sequence, one plan
            side = "in"
        attr = "_plan_in" if side == "in" else "_plan_out"
        if getattr(self, attr) is None:
            setattr(self, attr, ops.SplitPlan(self.slot_capacity, self.rowptr.device))
        return getattr(self, attr)

    def record_stream(self, stream, seen: Optional[set] = None) -> None:
        """Tell the caching allocator that `stream` uses this structure's tensors (built on another stream, e.g. by
        a loader that samples one batch ahead).  ``seen``: storages already recorded by the caller."""
        seen = set() if seen is None else seen  # a sampler batch keeps its arrays as views of one slab: one record
        for v in vars(self).values():
            if isinstance(v, Tensor) and v.is_cuda:
                key = v.untyped_storage().data_ptr()
                if key not in seen:
                    seen.add(key)
                    v.record_stream(stream)

    def in_degree(self) -> Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1]).long()

    def dropout_undirected(self, p: float, keep: Optional[Tensor] = None) -> "GraphStructure":
        """dropout_adj(edge_index, edge_attr, p, force_undirected=True) (reference pretrain.py:42-44)
        as a GraphStructure -> GraphStructure map on the device: no COO, no re-sort, no host
        sync.  The result's slots address THIS graph's edge_attr rows / edge types.  The Bernoulli
        draw is ``ops.dropout_keep_mask(E, p, *result.keep_key)`` unless ``keep`` is given."""
        self.ensure_transpose()
        out = GraphStructure(None, self.num_nodes)
        out.num_edges = self.num_edges
        out.slot_capacity = max(2 * self.num_edges, 1)  # every kept edge is mirrored: up to 2E live slots
        seed, offset = (0, 0) if keep is not None else ops.next_dropout_key()
        out.keep_key = (seed, offset)
        (out.rowptr, out.src, out.eid, out.etype_slot, out.dst_t, out.eid_t, out.etype_slot_t,
         out.inv_deg) = ops.graph_dropout_undirected(self, p, seed, offset, keep)
        out.rowptr_t = out.rowptr  # the augmented graph is symmetric: identical degree sequence
        # every kept edge has source id <= target id < active_rows (only row <= col survives, pretrain.py:42-44 /
        # dropout_adj), so after mirroring both endpoints of every edge are still below active_rows
        out.active_rows = self.active_rows
        if self.max_in_degree is not None and self.max_out_degree is not None:
            # a node keeps at most all of its in- and out-edges, each mirrored once
            out.max_in_degree = out.max_out_degree = self.max_in_degree + self.max_out_degree
        out._edge_type = self._edge_type
        return out


_CACHE: "OrderedDict[tuple, GraphStructure]" = OrderedDict()
_CACHE_SIZE = 8


def as_graph(edge_index, num_nodes: int, edge_type: Optional[Tensor] = None) -> GraphStructure:
    """GraphStructure for a raw edge_index tensor (cached by tensor identity + version)."""
    if isinstance(edge_index, GraphStructure):
        if edge_index.num_nodes != num_nodes:
            raise RuntimeError("GraphStructure was built for a different number of nodes")
        if edge_type is not None and not edge_index.has_edge_type():
            edge_index.set_edge_type(edge_type)
        return edge_index
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, int(num_nodes),
           None if edge_type is None else (edge_type.data_ptr(), edge_type._version))
    g = _CACHE.get(key)
    if g is not None and g._key_tensor is edge_index:
        _CACHE.move_to_end(key)
        return g
    g = GraphStructure(edge_index, num_nodes, edge_type)
    g._key_tensor = edge_index  # keeps the storage alive so data_ptr cannot be recycled under the key
    g._key_type = edge_type
    _CACHE[key] = g
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)
    return g


def clear_graph_cache() -> None:
    _CACHE.clear()
