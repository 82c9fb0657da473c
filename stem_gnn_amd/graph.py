"""Graph-structure object: the int32 CSR views of one int64 COO ``edge_index`` that the
HIP aggregation kernels consume, built once per (sub)graph on the device.

The reference has no such object: PyG's MessagePassing.propagate re-indexes the raw
``edge_index`` on every layer call (reference model/encoder.py:82).  ``GraphStructure`` may be
passed anywhere the modules accept ``edge_index``; a raw tensor is converted through a small
identity-keyed cache so a caller that only knows the reference API pays the build once per
distinct ``edge_index`` tensor.
"""
from __future__ import annotations

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


class EdgeTypeAttr:
    """Edge attributes given as (type table [T, D], type id per edge [E]) instead of the dense
    [E, D] tensor ``edge_text_feat[xe]`` the reference materialises on the host
    (reference pretrain.py:38).  ``dense()`` yields exactly that tensor."""

    def __init__(self, table: Tensor, etype: Tensor):
        if etype.dtype not in (torch.int32, torch.int64):
            raise RuntimeError("EdgeTypeAttr: etype must be an integer tensor")
        self.table = table.contiguous()
        self.etype = etype.contiguous()
        self._etype_i32 = None

    def etype_i32(self) -> Tensor:
        if self._etype_i32 is None:
            self._etype_i32 = self.etype.to(torch.int32)
        return self._etype_i32

    def size(self, dim: int) -> int:
        return (self.etype.numel(), self.table.size(1))[dim]

    def __getitem__(self, idx) -> "EdgeTypeAttr":
        return EdgeTypeAttr(self.table, self.etype[idx])

    def dense(self) -> Tensor:
        return self.table.index_select(0, self.etype.long())

    def to(self, device) -> "EdgeTypeAttr":
        return EdgeTypeAttr(self.table.to(device), self.etype.to(device))


class GraphStructure:
    """CSR grouped by target (forward) and, lazily, by source (backward)."""

    def __init__(self, edge_index: Tensor, num_nodes: int, edge_type: Optional[Tensor] = None,
                 validate: Optional[bool] = None):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError(f"edge_index: expected shape [2, E], got {tuple(edge_index.shape)}")
        self.edge_index = edge_index.contiguous()
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        self.rowptr, self.src, self.eid, self._bad = ops.csr_build(self.edge_index, self.num_nodes, 1)
        self.rowptr_t = self.dst_t = self.eid_t = self.inv_deg = None
        self.etype_slot = self.etype_slot_t = None
        self._edge_type = None
        if edge_type is not None:
            self.set_edge_type(edge_type)
        if _VALIDATE if validate is None else validate:
            bad = int(self._bad.item())
            if bad:
                raise IndexError(f"edge_index has {bad} entries outside [0, {self.num_nodes})")

    def set_edge_type(self, edge_type: Tensor) -> None:
        if edge_type.numel() != self.num_edges:
            raise RuntimeError("edge_type: one id per edge expected")
        self._edge_type = edge_type.to(torch.int32).contiguous()
        self.etype_slot = ops.gather_i32(self._edge_type, self.eid) if self.num_edges else self._edge_type
        self.etype_slot_t = None
        if self.eid_t is not None:
            self.etype_slot_t = ops.gather_i32(self._edge_type, self.eid_t) if self.num_edges else self._edge_type

    def ensure_transpose(self) -> None:
        if self.rowptr_t is None:
            self.rowptr_t, self.dst_t, self.eid_t, _ = ops.csr_build(self.edge_index, self.num_nodes, 0)
            self.inv_deg = ops.inv_degree(self.rowptr)
        if self._edge_type is not None and self.etype_slot_t is None:
            self.etype_slot_t = (ops.gather_i32(self._edge_type, self.eid_t) if self.num_edges else self._edge_type)

    def in_degree(self) -> Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1]).long()


_CACHE: "OrderedDict[tuple, GraphStructure]" = OrderedDict()
_CACHE_SIZE = 8


def as_graph(edge_index, num_nodes: int, edge_type: Optional[Tensor] = None) -> GraphStructure:
    """GraphStructure for a raw edge_index tensor (cached by tensor identity + version)."""
    if isinstance(edge_index, GraphStructure):
        if edge_index.num_nodes != num_nodes:
            raise RuntimeError("GraphStructure was built for a different number of nodes")
        if edge_type is not None and edge_index._edge_type is None:
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
