"""Encoder-side modules with the reference's class names, constructor signatures, forward
signatures and state-dict keys (reference STEM-GNN/model/encoder.py), computing through the
gfx950 HIP kernels.

    MySAGEConv            encoder.py:17-106
    MixtureSageLayer      encoder.py:109-129
    Encoder               encoder.py:132-333
    InnerProductDecoder   encoder.py:336-380

``edge_index`` may be the reference's int64 [2, E] tensor or a prebuilt
``stem_gnn_amd.graph.GraphStructure``; ``edge_attr`` may be the reference's dense [E, D]
tensor, ``None``, or an ``EdgeTypeAttr`` (type table + per-edge type id) which keeps the
[E, D] expansion out of HBM.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..graph import EdgeTypeAttr, GraphStructure, as_graph

EdgeAttr = Union[Tensor, EdgeTypeAttr, None]


def _pyg_linear_reset(lin: nn.Linear) -> None:
    """torch_geometric.nn.dense.linear.Linear.reset_parameters with the default initialisers
    (PyG 2.3.0): weight ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming_uniform, a=sqrt(5)),
    bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    bound = 1.0 / math.sqrt(lin.in_features) if lin.in_features > 0 else 0.0
    with torch.no_grad():
        lin.weight.uniform_(-bound, bound)
        if lin.bias is not None:
            lin.bias.uniform_(-bound, bound)


def _split_edge_attr(edge_attr: EdgeAttr):
    """-> (dense [E, D] or None, type table or None, type ids or None)"""
    if edge_attr is None:
        return None, None, None
    if isinstance(edge_attr, EdgeTypeAttr):
        return None, edge_attr.table, edge_attr.etype
    return edge_attr.contiguous(), None, None


def aggregate(x: Tensor, edge_index, edge_attr: EdgeAttr = None, num_nodes: Optional[int] = None) -> Tensor:
    """mean_{j->i} relu(x_j + xe_ji): MySAGEConv.propagate/message (encoder.py:82,94-97)."""
    n = x.size(0) if num_nodes is None else num_nodes
    dense, etab, etype = _split_edge_attr(edge_attr)
    graph = as_graph(edge_index, n, etype)
    if dense is not None and dense.requires_grad:
        raise NotImplementedError("edge_attr is data in the reference path; gradients w.r.t. it are not provided")
    return ops.SageAggFn.apply(x, graph, dense, etab)


class MySAGEConv(nn.Module):
    """out = lin_l(mean_{j->i} relu(x_j + xe_ji)) + lin_r(x_i)  (encoder.py:72-92).

    Only the configuration the reference instantiates is supported (encoder.py:193:
    aggr='mean', normalize=False, root_weight=True, project=False)."""

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, aggr: Optional[str] = "mean",
                 normalize: bool = False, root_weight: bool = True, project: bool = False, bias: bool = True,
                 **kwargs):
        super().__init__()
        if aggr != "mean":
            raise NotImplementedError("MySAGEConv: only aggr='mean' (the reference's setting) is implemented")
        if project:
            raise NotImplementedError("MySAGEConv: project=True is never used by the reference")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        self.root_weight = root_weight
        self.project = project
        self.aggr = aggr
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.lin_l = nn.Linear(in_channels[0], out_channels, bias=bias)
        if self.root_weight:
            self.lin_r = nn.Linear(in_channels[1], out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        _pyg_linear_reset(self.lin_l)
        if self.root_weight:
            _pyg_linear_reset(self.lin_r)

    def forward(self, x: Union[Tensor, Tuple[Tensor, Optional[Tensor]]], edge_index, edge_attr: EdgeAttr = None,
                size=None) -> Tensor:
        if isinstance(x, Tensor):
            x = (x, x)
        if size is not None and (size[0] != size[1] or size[0] != x[0].size(0)):
            raise NotImplementedError("bipartite propagation is not used by the reference path")
        out, _ = self._forward_fused(x, edge_index, edge_attr, want_stats=False)
        return out

    def _forward_fused(self, x, edge_index, edge_attr, want_stats: bool):
        """lin_l(agg) + lin_r(x) as ONE MFMA kernel over the concatenated K dimension; optionally
        also returns the BatchNorm column partials of the output (fused epilogue)."""
        agg = aggregate(x[0], edge_index, edge_attr)
        x_r = x[1]
        # a sampled batch promises that only its leading rows receive edges: the product skips the zero part of agg
        rows = getattr(edge_index, "active_rows", None) if isinstance(edge_index, GraphStructure) else None
        if self.root_weight and x_r is not None:
            out, partial = ops.LinearFn.apply(agg, self.lin_l.weight, x_r, self.lin_r.weight, self.lin_l.bias,
                                              want_stats and not self.normalize, -1 if rows is None else rows)
        else:
            out, partial = ops.LinearFn.apply(agg, self.lin_l.weight, None, None, self.lin_l.bias,
                                              want_stats and not self.normalize)
        if self.normalize:
            out = F.normalize(out, p=2.0, dim=-1)
            partial = None
        return out, partial

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, aggr={self.aggr})"


class MixtureSageLayer(nn.Module):
    """K-expert SAGE layer (encoder.py:109-129).  NOTE the reversed direction: row =
    edge_index[0] receives the mean of x[col] with no edge term and no relu (encoder.py:123-124),
    i.e. the mean aggregation over the transposed graph without edge attributes."""

    def __init__(self, in_dim: int, out_dim: int, num_experts: int, residual: bool = True):
        super().__init__()
        self.in_dim, self.out_dim, self.num_experts = in_dim, out_dim, num_experts
        self.residual = residual and (in_dim == out_dim)
        self.weights = nn.Parameter(torch.empty(num_experts, in_dim * 2, out_dim))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weights)

    def forward(self, x: Tensor, edge_index, edge_attr: EdgeAttr = None) -> Tensor:
        ei = edge_index.edge_index if isinstance(edge_index, GraphStructure) else edge_index
        flipped = _flipped_graph(ei, x.size(0))
        # scatter_mean(x[col], row) == plain mean aggregation over the flipped graph
        agg = ops.MeanAggFn.apply(x, flipped)
        combined = torch.cat([agg, x], dim=-1)
        k, d2, o = self.weights.shape
        if combined.is_cuda and d2 % 4 == 0 and (k * o) % 4 == 0:
            # einsum('nd,kdo->nko') as ONE product against the experts laid side by side [2D, K*O]
            wide = self.weights.permute(1, 0, 2).reshape(d2, k * o)
            outputs = ops.MatmulFn.apply(combined, wide).view(-1, k, o)
        else:
            outputs = torch.einsum("nd,kdo->nko", combined, self.weights)
        if self.residual:
            outputs = outputs + x.unsqueeze(1)
        return outputs


def _flipped_graph(edge_index: Tensor, num_nodes: int) -> GraphStructure:
    cache = getattr(edge_index, "_stemgnn_flipped", None)
    if cache is not None and cache[0] == edge_index._version and cache[1].num_nodes == num_nodes:
        return cache[1]
    g = GraphStructure(edge_index.flip(0).contiguous(), num_nodes)
    try:
        edge_index._stemgnn_flipped = (edge_index._version, g)
    except AttributeError:
        pass
    return g


class Encoder(nn.Module):
    """Layer stack + BatchNorm1d + activation + dropout (+ optional soft MoE routing)
    (encoder.py:132-333).  backbone='sage' only: the other backbones are stock PyG layers the
    reference never selects by default (config/pretrain.yaml:5)."""

    def __init__(self, input_dim, hidden_dim, activation, num_layers, backbone="sage", normalize="none",
                 dropout=0.0, moe=False, num_experts=3, tau=1.0, moe_layers="all"):
        super().__init__()
        if backbone != "sage":
            raise NotImplementedError(f"backbone={backbone!r}: only 'sage' is implemented on the HIP path")
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.backbone = backbone
        self.normalize = normalize
        self.moe = moe and num_experts > 1
        self.num_experts = num_experts
        self.tau = tau
        self.moe_layers = moe_layers

        self.activation = activation()
        if isinstance(self.activation, nn.ReLU):
            self._act_code, self._slope = 1, 0.0
        elif isinstance(self.activation, nn.LeakyReLU):
            self._act_code, self._slope = 1, float(self.activation.negative_slope)
        else:
            raise NotImplementedError("activation must be nn.ReLU or nn.LeakyReLU (reference pretrain.py:85)")
        self.layers = nn.ModuleList()
        self.norms = nn.ModuleList()
        self.dropout = nn.Dropout(dropout)
        self.env_encoders = nn.ModuleList()
        self._last_env_reg: Optional[Tensor] = None
        self._moe_usage: Optional[list] = None
        self._router_cache: Optional[list] = None
        self._cache_router = False
        self.last_dropout_keys: List[Tuple[int, int]] = []  # (seed, offset) per fused dropout of the last call
        self.gumbel_noise: Optional[list] = None    # injected Gumbel(0, 1) draws, one [N, experts] tensor per MoE layer
        self.last_gumbel_noise: List[Tensor] = []   # the draws the last training forward used

        self.moe_layer_flags = self._build_moe_layer_flags()
        dims = [input_dim] + [hidden_dim] * num_layers
        for layer_idx, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            if self.moe_layer_flags[layer_idx]:
                self.layers.append(MixtureSageLayer(in_dim, out_dim, self.num_experts, residual=True))
                self.env_encoders.append(nn.Linear(in_dim, self.num_experts))
            else:
                self.layers.append(self._build_conv(in_dim, out_dim))
            self.norms.append(nn.BatchNorm1d(out_dim))
        self.reset_parameters()

    def _build_moe_layer_flags(self):
        """Which layers are mixture layers: every one ('all'), the last one ('last') or none (encoder.py:139-156)."""
        n = self.num_layers
        if not self.moe or self.moe_layers == "none":
            return [False] * n
        if self.moe_layers not in ("all", "last"):
            raise ValueError(f"Unsupported moe_layers setting: {self.moe_layers}")
        return [self.moe_layers == "all" or i == n - 1 for i in range(n)]

    def _build_conv(self, in_dim, out_dim):
        return MySAGEConv(in_dim, out_dim, aggr="mean", normalize=False, root_weight=True)

    def _reg_loss(self, weights, logits):
        """E_nodes[sum_k w_k log softmax(logits)_k]: the routing weights' cross entropy against the router's own
        distribution, sign as in the reference (encoder.py:202-204)."""
        return (weights * F.log_softmax(logits, dim=-1)).sum(dim=-1).mean()

    def reset_parameters(self):
        for layer in self.layers:
            layer.reset_parameters()
        for norm in self.norms:
            norm.reset_parameters()
        for enc in self.env_encoders:
            enc.reset_parameters()
        self._last_env_reg = None
        self._moe_usage = None

    # -- router bookkeeping ------------------------------------------------------------------
    # Interface kept from the reference (encoder.py:219-277): enable_router_cache / get_router_cache hand back the
    # routing weights of every MoE layer seen since the last reset; get_moe_usage reports, per MoE layer, the mean
    # routing probability and the share of nodes whose top expert it was.
    def enable_router_cache(self, flag: bool = True):
        self._cache_router = bool(flag)
        self._router_cache = [] if flag else None

    def get_router_cache(self, reset: bool = True):
        cached = list(self._router_cache or ())
        if reset:
            self.enable_router_cache(self._cache_router)
        return cached

    def _update_moe_usage(self, env_idx, weights):
        """Accumulate [sum of probabilities | top-1 counts] per expert and the node count, on the device."""
        w = weights.detach()
        if self._moe_usage is None:
            self._moe_usage = [None] * sum(self.moe_layer_flags)
        top1 = torch.zeros_like(w).scatter_(1, w.argmax(dim=-1, keepdim=True), 1.0)
        tally = torch.stack([w.sum(dim=0), top1.sum(dim=0)])           # [2, experts]
        prev = self._moe_usage[env_idx]
        self._moe_usage[env_idx] = (tally, w.size(0)) if prev is None else (prev[0] + tally, prev[1] + w.size(0))

    def get_moe_usage(self, reset=True):
        report = []
        if self.moe and self._moe_usage is not None:
            moe_layers = [i for i, flag in enumerate(self.moe_layer_flags) if flag]
            for layer_idx, entry in zip(moe_layers, self._moe_usage):
                if entry is None:
                    continue
                tally, nodes = entry
                mean = (tally / max(nodes, 1)).cpu().tolist()
                report.append({"layer": layer_idx, "avg_prob": mean[0], "top1_frac": mean[1]})
            if reset:
                self._moe_usage = None
        return report

    # -- forward --------------------------------------------------------------------------
    def forward(self, x, edge_index, edge_attr=None):
        return self.encode(x, edge_index, edge_attr)

    def _norm_act_drop(self, i: int, z: Tensor, last: bool, partial: Optional[Tensor] = None) -> Tensor:
        """norms[i] -> activation -> dropout (encoder.py:313-317) as ONE fused op.  ``partial``:
        column partials of z from the producing linear kernel (saves the statistics pass)."""
        use_bn = self.normalize != "none"
        norm = self.norms[i]
        act = 0 if last else self._act_code
        p = 0.0 if (last or not self.training) else float(self.dropout.p)
        if use_bn and not self.training:
            # eval-mode BN uses running statistics: a per-column affine map, done with torch ops
            z = F.batch_norm(z, norm.running_mean, norm.running_var, norm.weight, norm.bias, False, 0.0, norm.eps)
            use_bn = False
        if not use_bn and act == 0 and p == 0.0:
            return z
        seed, offset = (0, 0)
        if p > 0.0:
            seed, offset = ops.next_dropout_key()
            self.last_dropout_keys.append((seed, offset))
        if use_bn:
            momentum = norm.momentum
            fused_stats = partial is not None and z.size(0) > 1
            nbt = None
            if norm.track_running_stats:
                if fused_stats and momentum is not None and norm.num_batches_tracked.is_cuda:
                    nbt = norm.num_batches_tracked  # bumped inside the statistics launch
                else:
                    norm.num_batches_tracked.add_(1)
                    if momentum is None:
                        momentum = 1.0 / float(norm.num_batches_tracked)
            rm = norm.running_mean if norm.track_running_stats else None
            rv = norm.running_var if norm.track_running_stats else None
            stats = None
            if fused_stats:
                stats = ops.bn_stats_from_partials(partial, partial.size(0), z.size(0), norm.eps, rm, rv, momentum, nbt)
            return ops.BnActDropFn.apply(z, norm.weight, norm.bias, rm, rv, True, momentum, norm.eps, act,
                                         self._slope, p, seed, offset, stats)
        return ops.BnActDropFn.apply(z, None, None, None, None, False, 0.0, 0.0, act, self._slope, p, seed, offset)

    def _encode_phase(self, x, graph, dense, etab, out_rows=None):
        """The whole layer stack as one library call per direction (ops.EncoderFn, csrc/phases.hip), or None when the
        configuration needs the per-layer path (MoE layers, cumulative-average BatchNorm, a graph with hub rows,
        gradients wanted in eval mode)."""
        if any(self.moe_layer_flags) or not ops.encoder_phase_ok(graph, x, dense, etab):
            return None
        use_bn = self.normalize != "none"
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if use_bn and not self.training and wants_grad:
            return None
        per_layer, params = [], []
        keys = []
        for i, (conv, norm) in enumerate(zip(self.layers, self.norms)):
            if conv.normalize or not conv.root_weight or conv.in_channels != x.size(1) and i == 0:
                return None
            if use_bn and (norm.momentum is None or not norm.affine or (not norm.track_running_stats and not self.training)):
                return None
            last = i == self.num_layers - 1
            key = (0, 0)
            if self.training and not last and self.dropout.p > 0:
                key = ops.next_dropout_key()
                keys.append(key)
            track = use_bn and norm.track_running_stats
            per_layer.append(dict(running_mean=norm.running_mean if track else None,
                                  running_var=norm.running_var if track else None,
                                  num_batches_tracked=norm.num_batches_tracked if (track and self.training) else None,
                                  eps=float(norm.eps), momentum=float(norm.momentum or 0.0), drop_key=key))
            params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight,
                       norm.weight if use_bn else None, norm.bias if use_bn else None]
        cfg = dict(use_bn=use_bn, training=self.training, act=self._act_code, slope=self._slope,
                   p=float(self.dropout.p) if self.training else 0.0,
                   out_rows=out_rows if (out_rows and not wants_grad) else 0, wants_grad=wants_grad)
        self.last_dropout_keys = keys
        return ops.EncoderFn.apply(x, graph, dense, etab, (cfg, per_layer), *params)

    def encode(self, x, edge_index, edge_attr=None, out_rows=None):
        """``out_rows`` (not part of the reference signature): a no-grad caller that reads only the leading rows of the
        output may say so -- the last layer's statistics still run over every row, its values are produced for those
        rows only, and the result has ``out_rows`` rows."""
        dense, etab, etype = _split_edge_attr(edge_attr)
        graph = as_graph(edge_index, x.size(0), etype)  # one structure build shared by all layers
        self._last_env_reg = None
        self.last_dropout_keys = []
        z = self._encode_phase(x, graph, dense, etab, out_rows)
        if z is not None:
            self._last_env_reg = self._zero_reg(z.device)
            return z
        if x.dtype != torch.float32:
            raise NotImplementedError("bf16-stored features run through the encoder phase only (sage backbone without "
                                      "MoE layers, no hub rows); this configuration needs fp32 features")
        z = self._encode_layers(x, graph, edge_attr)
        return z if out_rows is None else z[:out_rows]

    def _encode_layers(self, x, graph, edge_attr):
        """Layer by layer through the single-op autograd functions (every configuration)."""
        z = x
        env_idx = 0
        env_reg_total: Optional[Tensor] = None
        env_layers = 0
        self.last_gumbel_noise = []

        for i in range(self.num_layers):
            layer = self.layers[i]
            if isinstance(layer, MixtureSageLayer):
                logits = self.env_encoders[env_idx](z)
                if self.training:
                    # F.gumbel_softmax(logits, tau) (encoder.py:294) = softmax((logits + G) / tau), G ~ Gumbel(0, 1);
                    # the draw is kept (last_gumbel_noise) and can be injected (gumbel_noise) so that a parity test
                    # replays the same routing through the oracle
                    if self.gumbel_noise is not None:
                        gumbel = self.gumbel_noise[env_idx].to(logits)
                    else:
                        gumbel = -torch.empty_like(logits).exponential_().log()
                    self.last_gumbel_noise.append(gumbel)
                    weights = F.softmax((logits + gumbel) / self.tau, dim=-1)
                    reg = self._reg_loss(weights, logits)
                    env_reg_total = reg if env_reg_total is None else env_reg_total + reg
                    env_layers += 1
                else:
                    weights = F.softmax(logits, dim=-1)
                if self._cache_router:
                    if self._router_cache is None:
                        self._router_cache = []
                    self._router_cache.append(weights.detach())
                self._update_moe_usage(env_idx, weights)
                expert_outputs = layer(z, graph, edge_attr)
                z = torch.sum(weights.unsqueeze(-1) * expert_outputs, dim=1)
                env_idx += 1
                partial = None
            else:
                want = self.normalize != "none" and self.training
                z, partial = layer._forward_fused((z, z), graph, edge_attr, want_stats=want)
            z = self._norm_act_drop(i, z, last=(i == self.num_layers - 1), partial=partial)

        if env_reg_total is not None and self.training and env_layers > 0:
            self._last_env_reg = env_reg_total / env_layers
        else:
            self._last_env_reg = self._zero_reg(z.device)
        return z

    def _zero_reg(self, device):
        """A constant zeros(1) per device (encoder.py:319-322 builds one per call: a fill launch per forward)."""
        cache = self.__dict__.setdefault("_zero_reg_cache", {})
        t = cache.get(device)
        if t is None:
            t = cache[device] = torch.zeros(1, device=device)
        return t

    def get_env_reg(self, reset=True):
        if self._last_env_reg is None:
            reg = self._zero_reg(next(self.parameters()).device)
        else:
            reg = self._last_env_reg
        if reset:
            self._last_env_reg = None
        return reg


class InnerProductDecoder(nn.Module):
    """sigma(<lin(z)_u, lin(z)_v>) per edge (encoder.py:336-366)."""

    def __init__(self, hidden_dim=None, output_dim=None):
        super().__init__()
        self.proj_z = False
        if hidden_dim is not None:
            self.proj_z = True
            self.lin = nn.Linear(hidden_dim, output_dim)

    def forward(self, z: Tensor, edge_index: Tensor, sigmoid: bool = True) -> Tensor:
        z = ops.linear(z, self.lin) if self.proj_z else z
        value = ops.EdgeDotFn.apply(z, edge_index)
        return torch.sigmoid(value) if sigmoid else value

    def forward_all(self, z: Tensor, sigmoid: bool = True) -> Tensor:
        z = self.lin(z) if self.proj_z else z
        adj = torch.matmul(z, z.t())
        return torch.sigmoid(adj) if sigmoid else adj
