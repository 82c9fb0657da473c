"""Multi-dataset pretraining mix (`--pretrain_dataset all`, BASELINE config 5): several graphs pretrained on together.

What the reference does (STEM-GNN/dataset/process_datasets.py:146-198, config/pt_data.yaml, pretrain.py:144-153):
* every dataset keeps its own text tables; ``x`` / ``xe`` (row ids into them) are shifted by the rows of the datasets
  before it, the tables are concatenated (``preprocess_dataset_list``), and PyG's ``Batch.from_data_list`` stacks the
  graphs into ONE disconnected graph whose ``ptr`` holds the node offset of each member;
* every epoch the seed list is rebuilt with per-dataset weights (``get_train_node_idx``): a graph with weight w
  contributes each of its nodes floor(w) times plus a random frac(w) share of them once -- 5 x cora, 10 x FB15K237,
  a random 10 % of the two big molecule sets, ...;
* ``NeighborLoader`` then samples ``[10] * num_layers`` neighbourhoods of shuffled seed batches in the union graph.

Here the union lives on the device (int64 COO + both text tables), the weighted seed list is built with device ops, and
the batches come from the HIP sampler (data/sampler.py).  The real datasets cannot be materialised offline
(SURVEY.md 8c); ``synthetic_mix`` builds a stand-in with the same member structure.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

from .synthetic import SyntheticGraph, make_graph

# config/pt_data.yaml, entry "all" (dataset -> seed weight); sub-mixes are subsets with the same weights
PT_DATA_WEIGHTS: Dict[str, float] = {"cora": 5, "pubmed": 5, "arxiv": 5, "wikics": 5, "WN18RR": 5, "FB15K237": 10,
                                     "chemhiv": 1, "chemblpre": 0.1, "chempcba": 0.1}
MIXES: Dict[str, List[str]] = {
    "all": list(PT_DATA_WEIGHTS), "node": ["cora", "pubmed", "arxiv", "wikics"], "link": ["WN18RR", "FB15K237"],
    "graph": ["chemhiv", "chemblpre", "chempcba"], "citation": ["cora", "pubmed", "arxiv"],
}


def mix_weights(setting: str) -> Dict[str, float]:
    """Member datasets and weights of a ``--pretrain_dataset`` setting (a mix name, ``wo_<dataset>`` or one dataset)."""
    if setting in MIXES:
        names = MIXES[setting]
    elif setting.startswith("wo_") and setting[3:] in PT_DATA_WEIGHTS:
        names = [n for n in PT_DATA_WEIGHTS if n != setting[3:]]
    elif setting in PT_DATA_WEIGHTS:
        names = [setting]
    else:
        raise KeyError(f"unknown pretrain_dataset setting {setting!r}")
    return {n: PT_DATA_WEIGHTS[n] for n in names}


@dataclass
class GraphUnion:
    """Several graphs as one disconnected graph (``Batch.from_data_list`` of the reference's members)."""
    num_nodes: int
    edge_index: Tensor        # int64 [2, E] in union node ids
    xe: Tensor                # int64 [E] rows of the concatenated edge_text_feat
    x: Tensor                 # int64 [N] rows of the concatenated node_text_feat
    node_text_feat: Tensor
    edge_text_feat: Tensor
    ptr: Tensor               # int64 [G + 1] node offset of every member (host tensor)
    names: List[str]


def merge_graphs(graphs: Sequence[SyntheticGraph], names: Optional[Sequence[str]] = None) -> GraphUnion:
    """process_datasets.py:166-182: shift every member's ``x`` / ``xe`` past the text rows of the members before it,
    its ``edge_index`` past their nodes, and concatenate."""
    if not graphs:
        raise ValueError("merge_graphs: no graphs")
    node_off = text_off = type_off = 0
    eis, xes, xs, ntf, etf, ptr = [], [], [], [], [], [0]
    for g in graphs:
        eis.append(g.edge_index + node_off)
        xes.append(g.xe + type_off)
        xs.append(g.x + text_off)
        ntf.append(g.node_text_feat)
        etf.append(g.edge_text_feat)
        node_off += g.num_nodes
        text_off += g.node_text_feat.size(0)
        type_off += g.edge_text_feat.size(0)
        ptr.append(node_off)
    return GraphUnion(node_off, torch.cat(eis, dim=1).contiguous(), torch.cat(xes).contiguous(), torch.cat(xs).contiguous(),
                      torch.cat(ntf).contiguous(), torch.cat(etf).contiguous(), torch.tensor(ptr, dtype=torch.int64),
                      list(names) if names is not None else [f"graph{i}" for i in range(len(graphs))])


def get_train_node_idx(ptr: Tensor, weights: Sequence[float], device=None, generator: Optional[torch.Generator] = None) -> Tensor:
    """The weighted seed list of one epoch (process_datasets.py:186-198): member i contributes ``arange(ptr[i],
    ptr[i+1])`` repeated ``int(w_i)`` times, plus the first ``int(frac(w_i) * n_i)`` entries of a random permutation
    of its nodes.  Members in order; the caller shuffles (NeighborLoader(shuffle=True))."""
    bounds = ptr.tolist()
    if len(bounds) != len(weights) + 1:
        raise ValueError("get_train_node_idx: one weight per member graph expected")
    dev = torch.device(device) if device is not None else ptr.device
    parts = []
    for (s, e), w in zip(zip(bounds[:-1], bounds[1:]), weights):
        whole, frac = int(w), float(w) - int(w)
        arr = torch.arange(s, e, device=dev)
        if whole:
            parts.append(arr.repeat(whole))
        # one permutation per member whatever its weight, like the reference (the generator's stream stays aligned)
        perm = torch.randperm(e - s, device=dev, generator=generator)
        extra = int(frac * (e - s))
        if extra:
            parts.append(arr[perm[:extra]])
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=dev)


# ---- synthetic stand-in of the nine members: node / edge / text-row / edge-type counts of the real datasets where
# they fit a test budget, the two pre-training molecule sets scaled down (they enter with weight 0.1 anyway)
_MEMBERS = {
    # name: (nodes, directed edge entries, unique node texts (0 = one per node), edge types, block = molecule size)
    "cora": (2_708, 10_556, 0, 1, 0), "pubmed": (19_717, 88_648, 0, 1, 0), "arxiv": (169_343, 2_315_598, 0, 1, 0),
    "wikics": (11_701, 431_726, 0, 1, 0), "WN18RR": (40_943, 173_670, 0, 11, 0), "FB15K237": (14_541, 544_230, 0, 237, 0),
    "chemhiv": (1_049_162, 2_259_376, 119, 4, 26), "chemblpre": (1_200_000, 2_600_000, 119, 4, 26),
    "chempcba": (1_200_000, 2_600_000, 119, 4, 26),
}


def _molecule_like(num_nodes: int, num_edges: int, dim: int, types: int, text_rows: int, block: int, device,
                   seed: int) -> SyntheticGraph:
    """Many small components: both endpoints of an edge lie in the same block of ``block`` consecutive nodes; node texts
    are drawn from a small shared table (atom types)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    half = num_edges // 2
    u = torch.randint(0, num_nodes, (half,), generator=g, device=dev)
    v = (u // block) * block + torch.randint(0, block, (half,), generator=g, device=dev)
    v = torch.where(v >= num_nodes, u, v)
    v = torch.where(v == u, torch.where(u % block == 0, u + 1, u - 1).clamp(0, num_nodes - 1), v)
    ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])]).contiguous()
    xe_half = torch.randint(0, types, (half,), generator=g, device=dev)
    ntf = F.normalize(torch.randn(text_rows, dim, generator=g, device=dev), dim=-1)
    etf = F.normalize(torch.randn(types, dim, generator=g, device=dev), dim=-1)
    x = torch.randint(0, text_rows, (num_nodes,), generator=g, device=dev)
    return SyntheticGraph(num_nodes, ei, torch.cat([xe_half, xe_half]).contiguous(), x, ntf, etf)


def synthetic_mix(setting: str = "all", dim: int = 768, device="cpu", scale: float = 1.0, seed: int = 1234) -> GraphUnion:
    """Stand-in for ``get_pt_data(data_path, setting)``: one synthetic graph per member with the member's sizes
    (multiplied by ``scale``), merged like the reference merges the real ones."""
    graphs, names = [], []
    for i, name in enumerate(mix_weights(setting)):
        n, e, rows, types, block = _MEMBERS[name]
        n, e = max(int(n * scale), 64), max(int(e * scale) // 2 * 2, 64)
        if block:
            graphs.append(_molecule_like(n, e, dim, types, rows, block, device, seed + i))
        else:
            graphs.append(make_graph(n, e, dim, types, kind="U", device=device, graph_seed=seed + i, feat_seed=i,
                                     feat_rows=0 if rows == 0 else rows))
        names.append(name)
    return merge_graphs(graphs, names)
