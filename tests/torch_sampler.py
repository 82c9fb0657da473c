"""Test aid: the NeighborLoader sampling contract (reference pretrain.py:151-153) restated with torch index
arithmetic over a CSC built by the HIP graph builder -- a second, independent implementation the fused HIP sampler
(stem_gnn_amd.data.sampler.HipNeighborSampler) is compared with.  Not part of the product package."""
from typing import List

import torch
from torch import Tensor

from stem_gnn_amd import ops
from stem_gnn_amd.data.sampler import Batch


class NeighborSampler:
    def __init__(self, edge_index: Tensor, xe: Tensor, num_nodes: int, x: Tensor, node_text_feat: Tensor,
                 edge_text_feat: Tensor, num_neighbors: List[int], seed: int = 0):
        dev = edge_index.device
        self.num_nodes = num_nodes
        self.fanouts = list(num_neighbors)
        rowptr, src, eid, _ = ops.csr_build(edge_index.contiguous(), num_nodes, 1)  # in-neighbour lists
        self.rowptr = rowptr.long()
        self.src = src
        self.xe_csc = xe.to(torch.int32)[eid.long()].contiguous()
        self.x, self.ntf, self.etf = x, node_text_feat, edge_text_feat
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        self._local = torch.full((num_nodes,), -1, dtype=torch.int64, device=dev)  # global -> local scratch

    def _sample_hop(self, frontier: Tensor, fanout: int):
        """-> (dst_global [M], src_global [M], etype [M]) for the sampled in-edges of `frontier`."""
        dev = frontier.device
        start = self.rowptr[frontier]
        deg = self.rowptr[frontier + 1] - start
        total = int(deg.sum().item())
        if total == 0:
            e = torch.empty(0, dtype=torch.int64, device=dev)
            return e, e, e
        seg = torch.repeat_interleave(torch.arange(frontier.numel(), device=dev), deg)
        seg_start = torch.cumsum(deg, 0) - deg
        within = torch.arange(total, device=dev) - seg_start[seg]
        slot = start[seg] + within
        if fanout >= 0:
            key = torch.rand(total, generator=self.gen, device=dev, dtype=torch.float64)
            order = torch.argsort(seg.double() + key)  # random order inside each segment
            rank = torch.arange(total, device=dev) - seg_start[seg[order]]
            pick = order[rank < fanout]
            pick, _ = torch.sort(pick)  # keep CSC order among the chosen edges
            seg, slot = seg[pick], slot[pick]
        return frontier[seg], self.src[slot].long(), self.xe_csc[slot].long()

    def sample(self, seeds: Tensor) -> Batch:
        dev = seeds.device
        local = self._local
        nodes = [seeds]
        local[seeds] = torch.arange(seeds.numel(), device=dev)
        count = seeds.numel()
        frontier = seeds
        srcs, dsts, ets = [], [], []
        for fanout in self.fanouts:
            d, s, t = self._sample_hop(frontier, fanout)
            srcs.append(s); dsts.append(d); ets.append(t)
            new = torch.unique(s[local[s] < 0]) if s.numel() else s
            if new.numel():
                local[new] = torch.arange(count, count + new.numel(), device=dev)
                count += new.numel()
                nodes.append(new)
            frontier = new
            if frontier.numel() == 0:
                break
        n_id = torch.cat(nodes)
        src = torch.cat(srcs) if srcs else seeds.new_empty(0)
        dst = torch.cat(dsts) if dsts else seeds.new_empty(0)
        et = torch.cat(ets) if ets else seeds.new_empty(0)
        edge_index = torch.stack([local[src], local[dst]], dim=0).contiguous()
        local[n_id] = -1  # reset the scratch map
        return Batch(batch_size=seeds.numel(), n_id=n_id, x=self.x[n_id], edge_index=edge_index, xe=et,
                     node_text_feat=self.ntf, edge_text_feat=self.etf)
