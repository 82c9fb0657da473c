"""``get_loader`` of the reference (STEM-GNN/utils/loader.py) on the device-resident HIP sampler: the training and
evaluation loaders of the three finetune tasks, with the reference's fan-outs and batch sizes.

* node  (utils/loader.py:9-26):  NeighborLoader [10] * L over the training nodes (shuffled), and the full-neighbourhood
  loader [-1] * L over every node in batches of 512 -> ``(train_loader, subgraph_loader)``.
* link  (utils/loader.py:27-46):  LinkNeighborLoader [30] * L over the training edges (shuffled), and [-1] * L over every
  edge in batches of 4 096 -> ``(train_loader, subgraph_loader)``.
* graph (utils/loader.py:48-72):  one loader of disjoint-union batches per split -> ``(train, val, test)``.

``data``: an object with ``edge_index`` int64 [2, E], ``xe`` int64 [E], ``node_text_feat``, ``edge_text_feat`` and,
optionally, ``x`` (int64 rows of ``node_text_feat`` per node; default: the identity) -- node / link tasks; a sequence of
such small graphs with ``y`` -- graph task.  Everything is moved to ``device`` once; batches are built there."""
from typing import Optional, Sequence

import torch

from ..data.sampler import HipNeighborSampler, LinkNeighborLoader, NeighborLoader
from .others import mask2idx


class GraphBatch:
    """Disjoint union of small graphs with the attributes task/graph.py reads."""

    def __init__(self, node_text_feat, edge_index, edge_text_feat, batch, y):
        self.node_text_feat, self.edge_index, self.edge_text_feat, self.batch, self.y = (node_text_feat, edge_index,
                                                                                          edge_text_feat, batch, y)

    def to(self, device):
        return GraphBatch(*(t.to(device) for t in (self.node_text_feat, self.edge_index, self.edge_text_feat, self.batch,
                                                   self.y)))


class GraphDataLoader:
    """DataLoader over a list of small graphs (reference utils/loader.py:53-70): a batch is the disjoint union of
    ``batch_size`` graphs -- node rows and edge rows concatenated, edge endpoints shifted by the graphs' node offsets,
    ``batch[i]`` = the graph of node i, ``y`` stacked per graph.  A graph gives ``node_text_feat`` [n, D] (or the table
    and ``x`` row ids), ``edge_index``, ``edge_text_feat`` [e, D] (or the table and ``xe``) and ``y``."""

    def __init__(self, graphs: Sequence, batch_size: int, shuffle: bool = False, seed: int = 0, device=None):
        self.graphs, self.batch_size, self.shuffle, self.seed, self.device, self.epoch = (list(graphs), batch_size, shuffle,
                                                                                          seed, device, 0)

    def __len__(self):
        return (len(self.graphs) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(range(len(self.graphs)))
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(len(order), generator=g).tolist()
            self.epoch += 1
        for i in range(0, len(order), self.batch_size):
            part = [self.graphs[j] for j in order[i:i + self.batch_size]]
            feats, edges, attrs, owner, ys, base = [], [], [], [], [], 0
            for k, gph in enumerate(part):
                x, xe = getattr(gph, "x", None), getattr(gph, "xe", None)
                f = gph.node_text_feat[x] if x is not None and x.dtype == torch.int64 and x.dim() == 1 else gph.node_text_feat
                a = gph.edge_text_feat[xe] if xe is not None else gph.edge_text_feat
                feats.append(f)
                attrs.append(a)
                edges.append(gph.edge_index + base)
                owner.append(torch.full((f.size(0),), k, dtype=torch.int64, device=f.device))
                ys.append(gph.y.reshape(1, -1))
                base += f.size(0)
            b = GraphBatch(torch.cat(feats), torch.cat(edges, dim=1), torch.cat(attrs), torch.cat(owner), torch.cat(ys))
            yield b.to(self.device) if self.device is not None else b


def _sampler(data, fanouts, device, seed):
    x = getattr(data, "x", None)
    n = int(getattr(data, "num_nodes", 0) or (x.numel() if x is not None else data.node_text_feat.size(0)))
    x = torch.arange(n, device=device) if x is None else x.to(device)
    return HipNeighborSampler(data.edge_index.to(device), data.xe.to(device), n, x, data.node_text_feat.to(device),
                              data.edge_text_feat.to(device), fanouts, seed=seed)


def get_loader(data, split, labels, params, device: Optional[torch.device] = None, seed: int = 0):
    task = params["task"]
    assert params["setting"] == "standard", "Only standard setting is supported"
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    layers = params["num_layers"]
    if task == "node":
        train = _sampler(data, [10] * layers, device, seed)
        y = labels.to(device)
        train_loader = NeighborLoader(train, mask2idx(split["train"]).to(device), params["batch_size"], shuffle=True, seed=seed,
                                      y=y)
        subgraph_loader = NeighborLoader(train.with_fanouts([-1] * layers), torch.arange(train.num_nodes, device=device), 512,
                                         shuffle=False, y=y)
        return train_loader, subgraph_loader
    if task == "link":
        train = _sampler(data, [30] * layers, device, seed)
        ei, y = data.edge_index.to(device), labels.to(device)
        mask = split["train"].to(device)
        train_loader = LinkNeighborLoader(train, ei[:, mask], y[mask], params["batch_size"], shuffle=True, seed=seed)
        subgraph_loader = LinkNeighborLoader(train.with_fanouts([-1] * layers), ei, y, 4096, shuffle=False)
        return train_loader, subgraph_loader
    if task == "graph":
        def pick(name):  # a split is a boolean mask over the graphs or a list of their indices
            ids = split[name]
            ids = mask2idx(ids) if ids.dtype == torch.bool else ids
            return [data[int(i)] for i in ids.tolist()]

        return (GraphDataLoader(pick("train"), params["batch_size"], shuffle=True, seed=seed, device=device),
                GraphDataLoader(pick("valid"), params["batch_size"], device=device),
                GraphDataLoader(pick("test"), params["batch_size"], device=device))
    raise ValueError(f"unknown task {task!r}")
