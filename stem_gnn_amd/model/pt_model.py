"""PretrainModel with the reference's interface (reference STEM-GNN/model/pt_model.py:11-142):
encoder + vq + three decoders + EMA teacher (``sem_encoder``) + ``sem_projector``; the four
reconstruction losses, the commitment loss and the MoE regulariser.

Random draws inside ``forward`` (edge sub-sampling permutations, negative edges) are taken on
the device and recorded in ``self.last_draws`` so a parity test can replay exactly the same
step through the CPU oracle.  They can also be injected through ``draws=``.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..graph import EdgeTypeAttr, GraphStructure, as_graph
from .encoder import InnerProductDecoder

EPS = 1e-15  # pt_model.py:8


def _edge_index_of(ei) -> Tensor:
    return ei.edge_index if isinstance(ei, GraphStructure) else ei


def _flat_param_views(module: nn.Module):
    """Re-home every parameter of `module` as a view into one flat fp32 buffer (so the teacher
    EMA is a single kernel over contiguous memory).  Returns the flat buffer."""
    params = list(module.parameters())
    total = sum(p.numel() for p in params)
    if total == 0:
        return None
    flat = torch.empty(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p.data)
        off += n
    return flat


class PretrainModel(nn.Module):
    def __init__(self, encoder, vq, feat_recon_decoder, topo_recon_decoder, topo_sem_recon_decoder):
        super().__init__()
        self.encoder = encoder
        self.vq = vq
        self.feat_recon_decoder = feat_recon_decoder
        self.topo_recon_decoder = topo_recon_decoder
        self.topo_sem_recon_decoder = topo_sem_recon_decoder
        self.sem_encoder = deepcopy(self.encoder)  # pt_model.py:22
        self.sem_projector = nn.Linear(self.encoder.hidden_dim, self.encoder.hidden_dim)
        self._flat_student: Optional[Tensor] = None
        self._flat_teacher: Optional[Tensor] = None
        self.last_draws: Dict[str, Tensor] = {}

    @property
    def get_encoder(self):
        return self.encoder

    @property
    def get_vq(self):
        return self.vq

    def save_encoder(self, path):
        torch.save(self.encoder.state_dict(), path)

    def save_vq(self, path):
        torch.save(self.vq.state_dict(), path)

    # -- losses ---------------------------------------------------------------------------
    @staticmethod
    def _lin(module, t):
        return ops.linear(t, module) if isinstance(module, nn.Linear) else module(t)

    def feat_recon(self, z):
        return self._lin(self.feat_recon_decoder, z)

    def feat_recon_loss(self, z, x, bs=None):
        return F.mse_loss(self.feat_recon(z[:bs]), x[:bs])  # pt_model.py:42-43

    def _sample_edges(self, num_edges: int, ratio: float, device, key: str, draws) -> Optional[Tensor]:
        """randperm(E)[:max(int(E*ratio),1)] (pt_model.py:51-57, 72-78), drawn on the device."""
        if ratio == 1.0:
            return None
        if draws is not None and key in draws:
            perm = draws[key]
        else:
            # a random k-subset (what randperm(E)[:k] is): first k outputs of a keyed pseudo-random
            # permutation of [0, E) -- one kernel instead of a sort of E keys
            k = max(int(num_edges * ratio), 1)
            perm = ops.sample_subset(num_edges, k, device)
        self.last_draws[key] = perm
        return perm

    def topo_recon_loss(self, z, pos_edge_index, neg_edge_index=None, ratio=1.0, draws=None):
        if ratio == 0.0:
            return torch.tensor(0.0, device=z.device)
        graph = pos_edge_index if isinstance(pos_edge_index, GraphStructure) else None
        full_edge_index = _edge_index_of(pos_edge_index)
        pos_edge_index = full_edge_index
        num_edges = pos_edge_index.size(1)
        dec = self.topo_recon_decoder
        replay = draws is not None and ("topo_perm" in draws or "neg_edge_index" in draws)
        if (ratio != 1.0 and neg_edge_index is None and not replay and z.is_cuda and num_edges > 0
                and isinstance(dec, InnerProductDecoder)):
            # one launch draws the positives, gathers their endpoints into the left half of the [2, 2k] index
            # buffer and flags them; the negative sampler fills the right half (pt_model.py:53-60, no indexing ops)
            if graph is None:
                graph = as_graph(full_edge_index, z.size(0))
            k = max(int(num_edges * ratio), 1)
            perm, both, _, selected = ops.sample_edges(full_edge_index.contiguous(), None, k, want_selected=True,
                                                       pad_columns=k)
            seed, offset = ops.next_dropout_key()
            ops.negative_sample_into(graph, selected, k, seed, offset, both, k)
            self.last_draws["topo_perm"] = perm
            self.last_draws["neg_edge_index"] = both[:, k:]
            zl = ops.linear(z, dec.lin) if dec.proj_z else z
            return ops.EdgeBceLossFn.apply(zl, both, k)
        perm = self._sample_edges(num_edges, ratio, z.device, "topo_perm", draws)
        if perm is not None:
            pos_edge_index = pos_edge_index[:, perm]
        if neg_edge_index is None:
            if draws is not None and "neg_edge_index" in draws:
                neg_edge_index = draws["neg_edge_index"]
            else:
                # negative_sampling(pos_edge_index, N) (pt_model.py:60): negatives must avoid the
                # SAMPLED positives; membership is looked up through the graph's by-target CSR
                if graph is None:
                    graph = as_graph(full_edge_index, z.size(0))
                selected = torch.zeros(num_edges, dtype=torch.uint8, device=z.device)
                if perm is None:
                    selected.fill_(1)
                else:
                    selected[perm] = 1
                seed, offset = ops.next_dropout_key()
                neg_edge_index = ops.negative_sample(graph, selected, pos_edge_index.size(1), seed, offset)
        self.last_draws["neg_edge_index"] = neg_edge_index
        dec = self.topo_recon_decoder
        if isinstance(dec, InnerProductDecoder):
            # -log(sigmoid(<.,.>)+EPS) means over positives and negatives, fused (pt_model.py:62-65)
            zl = ops.linear(z, dec.lin) if dec.proj_z else z
            both = torch.cat([pos_edge_index, neg_edge_index], dim=1)
            return ops.EdgeBceLossFn.apply(zl, both, pos_edge_index.size(1))
        pos_loss = -torch.log(dec(z, pos_edge_index, sigmoid=True) + EPS).mean()
        neg_loss = -torch.log(1 - dec(z, neg_edge_index, sigmoid=True) + EPS).mean()
        return pos_loss + neg_loss

    def topo_sem_recon_loss(self, z, edge_index, edge_attr, ratio=1.0, draws=None):
        if ratio == 0.0:
            return torch.tensor(0.0, device=z.device)
        edge_index = _edge_index_of(edge_index)
        perm = self._sample_edges(edge_index.size(1), ratio, z.device, "topo_sem_perm", draws)
        if perm is not None:
            edge_index = edge_index[:, perm]
            edge_attr = edge_attr[perm]
        target = edge_attr.dense() if isinstance(edge_attr, EdgeTypeAttr) else edge_attr
        zz = ops.EdgeConcatFn.apply(z, edge_index.contiguous())  # cat([z[u], z[v]]), pt_model.py:80
        return F.mse_loss(self._lin(self.topo_sem_recon_decoder, zz), target)

    def _teacher_forward(self, g, rows=None):
        """sem_encoder(orig graph).detach() (pt_model.py:93); only its rows [:bs] are ever read (:96-97), which
        ``rows`` tells the encoder.  (Issuing it on a side HIP stream was measured in round 2: no gain, K1 slower
        when overlapped -- the step runs on one stream.)"""
        orig_x, orig_edge_index, orig_edge_attr = g[0], g[1], g[2]
        with torch.no_grad():
            if rows is not None and hasattr(self.sem_encoder, "_encode_phase"):
                return self.sem_encoder.encode(orig_x, orig_edge_index, orig_edge_attr, out_rows=rows)
            return self.sem_encoder(orig_x, orig_edge_index, orig_edge_attr)

    def sem_recon_loss(self, g, quantize, eta=1.0, bs=None, z_t=None):
        z = z_t if z_t is not None else self._teacher_forward(g, rows=None if bs is None else int(bs))
        # the projector is row-wise and only rows [:bs] are used (pt_model.py:94-97): project those rows only
        h = self._lin(self.sem_projector, quantize[:bs])
        if eta == 1.0 and h.is_cuda:
            return ops.CosineLossFn.apply(z[:bs], h)  # mean(1 - <z/|z|, h/|h|>), fused forward + backward
        z = F.normalize(z[:bs], dim=-1, p=2)
        h = F.normalize(h, dim=-1, p=2)
        loss = (1 - (z * h).sum(dim=-1)).pow_(eta)
        return loss.mean()

    @torch.no_grad()
    def ema_update_sem_encoder(self, decay=0.99):
        """param_k = param_k * decay + param_q * (1 - decay) over encoder parameters
        (pt_model.py:104-106), as one kernel over flat parameter buffers."""
        if self._flat_student is None or not self._flat_ok():
            self._flat_student = _flat_param_views(self.encoder)
            self._flat_teacher = _flat_param_views(self.sem_encoder)
        if self._flat_student is not None:
            ops.ema_lerp_(self._flat_teacher, self._flat_student, decay)

    def _flat_ok(self) -> bool:
        """The flat views are stale if someone re-assigned .data (e.g. .to(device), load_state_dict
        keeps them valid because it copies in place)."""
        p = next(self.encoder.parameters(), None)
        return p is None or (p.data_ptr() == self._flat_student.data_ptr() and p.device == self._flat_student.device)

    def encode(self, x, edge_index, edge_attr=None):
        return self.encoder(x, edge_index, edge_attr)

    def quantize(self, x, edge_index, edge_attr=None):
        z = self.encoder(x, edge_index, edge_attr)
        skip = getattr(self.vq, "skip_codes", None)
        if skip is not None:
            self.vq.skip_codes = True  # the fourth output (per-head codes) is discarded right here, pt_model.py:113
        try:
            quantize, indices, commit_loss, _ = self.vq(z)
        finally:
            if skip is not None:
                self.vq.skip_codes = skip
        return z, quantize, indices, commit_loss

    def _heads_phase(self, query, g, ratio, bs, draws, z_t=None):
        """The four heads as one library call per direction (ops.HeadsFn), or None when the call is not the standard
        pretraining configuration (pretrain.py:91-130: Linear decoders, InnerProductDecoder with its projection, edge
        attributes as (type table, int64 type ids), a sampling ratio in (0, 1), no injected draws)."""
        orig_x, orig_edge_index, orig_edge_attr = g[0], g[1], g[2]
        dec = self.topo_recon_decoder
        if not (query.is_cuda and query.dtype == torch.float32 and 0.0 < ratio < 1.0 and bs is not None and not draws
                and isinstance(orig_edge_attr, EdgeTypeAttr) and orig_edge_attr.etype is not None
                and orig_edge_attr.etype.dtype == torch.int64
                and isinstance(self.feat_recon_decoder, nn.Linear) and isinstance(self.topo_sem_recon_decoder, nn.Linear)
                and isinstance(dec, InnerProductDecoder) and dec.proj_z and query.size(0) >= 2
                and 0 < bs <= query.size(0) and orig_x.dtype in (torch.float32, torch.bfloat16)
                and orig_x.size(1) % 4 == 0):
            return None
        d = query.size(1)
        if (tuple(self.topo_sem_recon_decoder.weight.shape) != (d, 2 * d) or tuple(dec.lin.weight.shape) != (d, d)
                or tuple(self.feat_recon_decoder.weight.shape) != (orig_x.size(1), d)
                or tuple(self.sem_projector.weight.shape) != (d, d) or orig_edge_attr.table.size(1) != d):
            return None
        graph = orig_edge_index if isinstance(orig_edge_index, GraphStructure) else as_graph(orig_edge_index, query.size(0))
        num_edges = graph.edge_index.size(1)
        if num_edges == 0:
            return None
        if z_t is None:
            z_t = self._teacher_forward(g, rows=int(bs))
        k = max(int(num_edges * ratio), 1)
        seed, o1 = ops.next_dropout_key()
        _, o2 = ops.next_dropout_key()
        _, o3 = ops.next_dropout_key()
        lins = (self.feat_recon_decoder, dec.lin, self.topo_sem_recon_decoder, self.sem_projector)
        params = [t for lin in lins for t in (lin.weight, lin.bias)]
        # the feature head's target: the seed rows in fp32 (bf16-stored features widen exactly)
        x_feat = orig_x if orig_x.dtype == torch.float32 else orig_x[:int(bs)].float()
        return ops.HeadsFn.apply(query, graph, orig_edge_attr.table, orig_edge_attr.etype.contiguous(),
                                 x_feat.contiguous(), z_t.contiguous(), int(bs), k, (seed, (o1, o2, o3)),
                                 self.last_draws, *params)

    def forward(self, aug_g, g, topo_recon_ratio=1.0, bs=None, no_codebook=False, draws=None):
        x, edge_index, edge_attr = aug_g[0], aug_g[1], aug_g[2]
        orig_x, orig_edge_index, orig_edge_attr = g[0], g[1], g[2]
        self.last_draws = {}
        # The EMA teacher first (pt_model.py:93: it depends on nothing the student computes).  The caller's augmentation
        # has just read the original features, so the teacher's first aggregation and product find them in the Infinity
        # Cache instead of behind the student's ~350 MB of traffic (its K1 launch: 19 -> 13 us on a C4 batch).
        z_t = self._teacher_forward(g, rows=None if bs is None else int(bs)) if orig_x.is_cuda else None
        z, quantize, indices, commit_loss = self.quantize(x, edge_index, edge_attr)
        env_reg_loss = self.encoder.get_env_reg()
        if no_codebook:
            query = z
            commit_loss = torch.tensor(0.0, device=z.device)
        else:
            query = quantize
        fused = self._heads_phase(query, g, topo_recon_ratio, bs, draws, z_t)
        if fused is not None:
            feat_recon_loss, topo_recon_loss, topo_sem_recon_loss, sem_recon_loss = fused
        elif topo_recon_ratio not in (0.0, 1.0) and query.is_cuda:
            # fused fan-out of the query: one dense gradient buffer for its three consumers
            full_ei = _edge_index_of(orig_edge_index)
            typed = isinstance(orig_edge_attr, EdgeTypeAttr)
            fused = (typed and orig_edge_attr.etype is not None and orig_edge_attr.etype.dtype == torch.int64
                     and not (draws is not None and "topo_sem_perm" in draws) and full_ei.size(1) > 0)
            if fused:
                # picks, their endpoints and their edge types in one launch; the target rows in one gather
                k = max(int(full_ei.size(1) * topo_recon_ratio), 1)
                perm, sem_ei, sel_type, _ = ops.sample_edges(full_ei.contiguous(), orig_edge_attr.etype, k)
                self.last_draws["topo_sem_perm"] = perm
                target = ops.gather_rows(orig_edge_attr.table, sel_type, validate=False)  # ids the graph build already saw
            else:
                perm = self._sample_edges(full_ei.size(1), topo_recon_ratio, z.device, "topo_sem_perm", draws)
                sem_ei = full_ei[:, perm]
                attr = orig_edge_attr[perm]
                target = attr.dense() if typed else attr
            q_all, q_head, zz = ops.QueryFanOutFn.apply(query, query.size(0) if bs is None else bs, sem_ei)
            feat_recon_loss = ops.MseLossFn.apply(self.feat_recon(q_head), orig_x[:bs])  # pt_model.py:42-43
            topo_recon_loss = self.topo_recon_loss(q_all, orig_edge_index, ratio=topo_recon_ratio, draws=draws)
            topo_sem_recon_loss = ops.MseLossFn.apply(self._lin(self.topo_sem_recon_decoder, zz), target)  # :80-81
            sem_recon_loss = self.sem_recon_loss(g, q_head, eta=1.0, bs=bs, z_t=z_t)
        else:
            feat_recon_loss = self.feat_recon_loss(query, orig_x, bs=bs)
            topo_recon_loss = self.topo_recon_loss(query, orig_edge_index, ratio=topo_recon_ratio, draws=draws)
            topo_sem_recon_loss = self.topo_sem_recon_loss(query, orig_edge_index, orig_edge_attr,
                                                           ratio=topo_recon_ratio, draws=draws)
            sem_recon_loss = self.sem_recon_loss(g, query, eta=1.0, bs=bs, z_t=z_t)
        losses = {
            "feat_recon_loss": feat_recon_loss,
            "topo_recon_loss": topo_recon_loss,
            "topo_sem_recon_loss": topo_sem_recon_loss,
            "sem_recon_loss": sem_recon_loss,
            "commit_loss": commit_loss,
            "env_reg_loss": env_reg_loss,
        }
        return z, quantize, indices, losses
