"""Synthetic pretraining graphs of SURVEY.md §8d / BASELINE.md §3 (no dataset can be
materialised offline): Graph-U (uniform undirected pairs, mirrored, shuffled) and Graph-Z
(Zipf-skewed targets), unit-norm Gaussian node features, T unit-norm edge-type rows."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class SyntheticGraph:
    num_nodes: int
    edge_index: Tensor        # int64 [2, E], unsorted COO, row 0 = source, row 1 = target
    xe: Tensor                # int64 [E] edge-type id per edge
    x: Tensor                 # int64 [N] node -> row of node_text_feat (identity here)
    node_text_feat: Tensor    # fp32 [N or feat_rows, D], unit-norm rows
    edge_text_feat: Tensor    # fp32 [T, D], unit-norm rows


def make_graph(num_nodes: int, num_edges: int, dim: int, num_edge_types: int = 4, kind: str = "U",
               device="cpu", graph_seed: int = 1234, feat_seed: int = 0, feat_rows: int = 0) -> SyntheticGraph:
    """``feat_rows`` > 0: a text table of that many unique rows with ``x`` drawn uniformly into it (the
    multi-dataset mixes, where many nodes share a text, e.g. atoms); 0: one row per node, ``x`` the identity."""
    if num_edges % 2:
        raise ValueError("num_edges must be even (mirrored undirected pairs)")
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(graph_seed)
    half = num_edges // 2
    u = torch.randint(0, num_nodes, (half,), generator=g, device=dev)
    if kind == "U":
        v = torch.randint(0, num_nodes - 1, (half,), generator=g, device=dev)
        v = v + (v >= u).long()  # u != v, uniform over the other nodes
    elif kind == "Z":
        # Zipf(alpha=1) over a random node permutation: P(rank r) ~ 1/r, via inverse CDF of 1/x
        r = torch.rand(half, generator=g, device=dev, dtype=torch.float64)
        rank = torch.exp(r * torch.log(torch.tensor(float(num_nodes), dtype=torch.float64, device=dev))).long() - 1
        rank.clamp_(0, num_nodes - 1)
        perm = torch.randperm(num_nodes, generator=g, device=dev)
        v = perm[rank]
        v = torch.where(v == u, (v + 1) % num_nodes, v)
    else:
        raise ValueError(kind)
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])], dim=0)
    shuffle = torch.randperm(num_edges, generator=g, device=dev)
    ei = ei[:, shuffle].contiguous()
    xe_half = torch.randint(0, num_edge_types, (half,), generator=g, device=dev)
    xe = torch.cat([xe_half, xe_half])[shuffle].contiguous()  # both directions of a pair share the type
    gf = torch.Generator(device=dev).manual_seed(feat_seed)
    rows = feat_rows if feat_rows > 0 else num_nodes
    ntf = F.normalize(torch.randn(rows, dim, generator=gf, device=dev), dim=-1)
    etf = F.normalize(torch.randn(num_edge_types, dim, generator=gf, device=dev), dim=-1)
    x = (torch.randint(0, rows, (num_nodes,), generator=gf, device=dev) if feat_rows > 0
         else torch.arange(num_nodes, device=dev))
    return SyntheticGraph(num_nodes, ei, xe, x, ntf, etf)
