"""Device-side equivalents of the torch_geometric.utils functions the pretraining loop calls
(reference pretrain.py:41-44, model/pt_model.py:60), following PyG 2.3.0's semantics
(reference environment.yml:292).  They run on whatever device the inputs live on with plain
torch ops (index arithmetic, no floating-point compute); each returns the random draw it used
so a parity test can replay the step through the CPU oracle.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from ..graph import EdgeTypeAttr, GraphStructure


def mask_feature(x: Tensor, p: float = 0.5, mode: str = "col", fill_value: float = 0.0,
                 training: bool = True, keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """PyG mask_feature: mode='col' draws ONE keep mask per feature column shared by all rows
    (``rand(1, D) >= p``); masked entries become ``fill_value``.  Returns (x_masked, mask)."""
    if p < 0.0 or p > 1.0:
        raise ValueError(f"Masking ratio has to be between 0 and 1 (got {p})")
    if not training or p == 0.0:
        return x, torch.ones_like(x, dtype=torch.bool)
    if mode == "row":
        mask = (torch.rand(x.size(0), device=x.device) >= p) if keep is None else keep
        mask = mask.view(-1, 1)
    elif mode == "col":
        mask = (torch.rand(x.size(1), device=x.device) >= p) if keep is None else keep
        mask = mask.view(1, -1)
    elif mode == "all":
        mask = (torch.rand_like(x) >= p) if keep is None else keep
    else:
        raise ValueError(f"unknown mode {mode!r}")
    return x.masked_fill(~mask, fill_value), mask


def dropout_adj(edge_index, edge_attr=None, p: float = 0.5, force_undirected: bool = False,
                num_nodes: Optional[int] = None, training: bool = True, keep: Optional[Tensor] = None):
    """PyG dropout_adj.  With force_undirected=True: entries with row > col are dropped first,
    the rest survive with probability 1-p, and each survivor is emitted in both directions
    ([row; col] then [col; row]) with its edge_attr duplicated.  edge_attr may be a dense
    tensor, an EdgeTypeAttr or None.  Returns (edge_index, edge_attr); the Bernoulli draw
    (before the row > col filter) is left in ``dropout_adj.last_keep``."""
    if p < 0.0 or p > 1.0:
        raise ValueError(f"Dropout probability has to be between 0 and 1 (got {p})")
    ei = edge_index.edge_index if isinstance(edge_index, GraphStructure) else edge_index
    if not training or p == 0.0:
        return ei, edge_attr
    row, col = ei[0], ei[1]
    mask = (torch.rand(row.size(0), device=ei.device) >= p) if keep is None else keep.clone()
    dropout_adj.last_keep = mask.clone()
    if force_undirected:
        mask = mask & ~(row > col)
    sel = mask.nonzero(as_tuple=False).view(-1)
    row, col = row[sel], col[sel]
    if edge_attr is not None:
        edge_attr = edge_attr[sel]
    if force_undirected:
        out = torch.stack([torch.cat([row, col], dim=0), torch.cat([col, row], dim=0)], dim=0)
        if isinstance(edge_attr, EdgeTypeAttr):
            edge_attr = EdgeTypeAttr(edge_attr.table, torch.cat([edge_attr.etype, edge_attr.etype], dim=0))
        elif edge_attr is not None:
            edge_attr = torch.cat([edge_attr, edge_attr], dim=0)
    else:
        out = torch.stack([row, col], dim=0)
    return out, edge_attr


dropout_adj.last_keep = None


def negative_sampling(edge_index: Tensor, num_nodes: int, num_neg_samples: Optional[int] = None) -> Tensor:
    """PyG negative_sampling (structured, method='sparse', not bipartite, directed): draws from
    the N*(N-1) non-self-loop pairs, rejects positives (up to three 1.1x-oversampled tries) and
    returns up to ``num_neg_samples`` (default: #positive edges) pairs.  The reference draws each
    try without replacement with Python's ``random.sample``; here the draw is ``randint`` on the
    device, so duplicate negatives are possible with probability ~k^2 / (2 N^2)."""
    n = int(num_nodes)
    dev = edge_index.device
    row, col = edge_index[0], edge_index[1]
    nonloop = row != col
    r, c = row[nonloop], col[nonloop]
    c = torch.where(r < c, c - 1, c)
    idx = r * (n - 1) + c  # edge_index_to_vector
    population = n * n - n
    k = edge_index.size(1) if num_neg_samples is None else num_neg_samples
    k = min(k, population - idx.numel())
    if k <= 0 or population <= 0:
        return edge_index.new_empty((2, 0))
    prob = 1.0 - idx.numel() / population
    sample_size = int(1.1 * k / prob)
    neg_idx = None
    for _ in range(3):
        if population <= sample_size:
            rnd = torch.arange(population, device=dev)
        else:
            rnd = torch.randint(0, population, (sample_size,), device=dev)
        rnd = rnd[~torch.isin(rnd, idx)]
        neg_idx = rnd if neg_idx is None else torch.cat([neg_idx, rnd])
        if neg_idx.numel() >= k:
            neg_idx = neg_idx[:k]
            break
    rr = torch.div(neg_idx, n - 1, rounding_mode="floor")
    cc = neg_idx % (n - 1)
    cc = torch.where(rr <= cc, cc + 1, cc)  # vector_to_edge_index
    return torch.stack([rr, cc], dim=0)
