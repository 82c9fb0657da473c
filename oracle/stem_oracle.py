"""CPU oracle for the STEM-GNN encoder + vector-quantize pretraining hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``stem_gnn_amd/`` imports, calls or links it; the
product path fails loudly when the HIP library is missing.

It is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm, written
as explicit index/scatter arithmetic so every line can be checked against the
reference by eye.  All ``file:line`` citations are relative to
``/root/reference/STEM-GNN``.

Pinning status
--------------
* Vector quantiser (``OracleVectorQuantize``): PINNED.  Checked against golden
  vectors produced by importing the reference's ``model/vq.py`` in the build
  container (``tests/golden/gen_vq_golden.py`` -> ``tests/golden/vq_*.pt``,
  ``tests/test_oracle_vq_golden.py``).
* Encoder / decoders / PretrainModel losses / PyG utilities: PARITY UNPINNED at
  the PyG boundary.  ``model/encoder.py`` and ``model/pt_model.py`` import
  ``torch_geometric`` / ``torch_scatter`` which are not installed (and cannot be
  installed), and the reference ships no tests or golden vectors.  The
  restatement follows the reference source plus PyG 2.3.0's documented semantics
  (environment.yml:292) and is pinned only by hand-derived known-answer tests
  (``tests/test_oracle_kat.py``).

Random draws (feature mask, edge drop, dropout masks, edge sub-sampling
permutations, negative edges, orthogonal-loss code ids) are *inputs* here: the
oracle never generates randomness inside the path, so it can be fed the exact
draws the HIP path used.
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

EPS = 1e-15  # model/pt_model.py:8


# --------------------------------------------------------------------------------------
# K1 / a1: MySAGEConv message + mean aggregation (model/encoder.py:72-97 through
# PyG 2.3.0 MessagePassing.propagate, flow source_to_target: j = edge_index[0],
# i = edge_index[1]; MeanAggregation = scatter-add of values and of ones,
# count.clamp(min=1), divide).
# --------------------------------------------------------------------------------------
def sage_mean_aggregate(x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor]) -> Tensor:
    """agg[i] = (1 / max(indeg(i), 1)) * sum_{e: dst(e)=i} relu(x[src(e)] + edge_attr[e])."""
    n, d = x.shape
    src, dst = edge_index[0], edge_index[1]
    msg = x.index_select(0, src)  # x_j (PyG gather along edge_index[0])
    if edge_attr is not None:  # encoder.py:95-96
        msg = msg + edge_attr
    msg = torch.relu(msg)  # encoder.py:97
    total = torch.zeros(n, d, dtype=x.dtype).index_add_(0, dst, msg)
    count = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
    return total / count.clamp(min=1).unsqueeze(1)


def scatter_mean_rows(src_rows: Tensor, index: Tensor, n: int) -> Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0, dim_size=n) for float input
    (sum / clamp(count, min=1)); used by MixtureSageLayer (encoder.py:124)."""
    total = torch.zeros(n, src_rows.size(1), dtype=src_rows.dtype).index_add_(0, index, src_rows)
    count = torch.zeros(n, dtype=src_rows.dtype).index_add_(
        0, index, torch.ones(index.numel(), dtype=src_rows.dtype))
    return total / count.clamp(min=1).unsqueeze(1)


class OracleSAGEConv(nn.Module):
    """MySAGEConv(in, out, aggr='mean', normalize=False, root_weight=True)
    (encoder.py:17-92).  lin_l has a bias, lin_r has none (encoder.py:58-60)."""

    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.lin_l = nn.Linear(in_dim, out_dim, bias=True)
        self.lin_r = nn.Linear(in_dim, out_dim, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor] = None) -> Tensor:
        agg = sage_mean_aggregate(x, edge_index, edge_attr)  # encoder.py:82
        return self.lin_l(agg) + self.lin_r(x)  # encoder.py:83-87


class OracleMixtureSageLayer(nn.Module):
    """MixtureSageLayer (encoder.py:109-129).  Direction is REVERSED relative to
    MySAGEConv: row = edge_index[0] receives the mean of x[col], no edge_attr and
    no relu (encoder.py:123-124)."""

    def __init__(self, in_dim: int, out_dim: int, num_experts: int):
        super().__init__()
        self.residual = in_dim == out_dim
        self.weights = nn.Parameter(torch.empty(num_experts, in_dim * 2, out_dim))
        nn.init.xavier_uniform_(self.weights)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr=None) -> Tensor:
        row, col = edge_index[0], edge_index[1]
        agg = scatter_mean_rows(x.index_select(0, col), row, x.size(0))
        combined = torch.cat([agg, x], dim=-1)
        out = torch.einsum('nd,kdo->nko', combined, self.weights)
        if self.residual:
            out = out + x.unsqueeze(1)
        return out


class OracleEncoder(nn.Module):
    """Encoder (encoder.py:132-333) for backbone='sage'.

    ``dropout_masks[i]`` is the boolean KEEP mask of layer i's dropout (layers
    0..L-2); kept elements are scaled by 1/(1-p) like nn.Dropout.  With
    ``dropout_masks=None`` dropout is the identity (p=0 or eval).
    ``gumbel_noise[k]`` is the Gumbel(0,1) sample added to the k-th MoE router's
    logits (F.gumbel_softmax, encoder.py:294).
    State-dict keys match the reference: layers.{i}.lin_l.weight/bias,
    layers.{i}.lin_r.weight, norms.{i}.*, env_encoders.{k}.*, layers.{i}.weights.
    """

    def __init__(self, input_dim: int, hidden_dim: int, num_layers: int, negative_slope: float = 0.0,
                 normalize: str = 'none', dropout: float = 0.0, moe: bool = False, num_experts: int = 3,
                 tau: float = 1.0, moe_layers: str = 'all'):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.normalize = normalize
        self.p = dropout
        self.negative_slope = negative_slope  # 0 -> nn.ReLU, 0.01 -> nn.LeakyReLU (pretrain.py:85)
        self.tau = tau
        self.moe = moe and num_experts > 1
        if not self.moe or moe_layers == 'none':
            flags = [False] * num_layers
        elif moe_layers == 'all':
            flags = [True] * num_layers
        elif moe_layers == 'last':
            flags = [False] * (num_layers - 1) + [True]
        else:
            raise ValueError(moe_layers)
        self.moe_layer_flags = flags
        dims = [input_dim] + [hidden_dim] * num_layers
        self.layers = nn.ModuleList()
        self.norms = nn.ModuleList()
        self.env_encoders = nn.ModuleList()
        for i in range(num_layers):
            if flags[i]:
                self.layers.append(OracleMixtureSageLayer(dims[i], dims[i + 1], num_experts))
                self.env_encoders.append(nn.Linear(dims[i], num_experts))
            else:
                self.layers.append(OracleSAGEConv(dims[i], dims[i + 1]))
            self.norms.append(nn.BatchNorm1d(dims[i + 1]))  # encoder.py:173 (also for normalize='layer')
        self._last_env_reg: Optional[Tensor] = None

    def _act(self, z: Tensor) -> Tensor:
        return F.leaky_relu(z, self.negative_slope) if self.negative_slope > 0 else torch.relu(z)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                dropout_masks: Optional[Sequence[Tensor]] = None,
                gumbel_noise: Optional[Sequence[Tensor]] = None) -> Tensor:
        z = x
        env_idx = 0
        reg_total = None
        for i in range(self.num_layers):
            layer = self.layers[i]
            if isinstance(layer, OracleMixtureSageLayer):
                logits = self.env_encoders[env_idx](z)
                if self.training:  # encoder.py:293-297
                    g = gumbel_noise[env_idx] if gumbel_noise is not None else torch.zeros_like(logits)
                    weights = torch.softmax((logits + g) / self.tau, dim=-1)
                    log_pi = logits - torch.logsumexp(logits, dim=-1, keepdim=True)  # encoder.py:202-204
                    reg = torch.mean(torch.sum(weights * log_pi, dim=-1))
                    reg_total = reg if reg_total is None else reg_total + reg
                else:
                    weights = torch.softmax(logits, dim=-1)
                z = torch.sum(weights.unsqueeze(-1) * layer(z, edge_index, edge_attr), dim=1)
                env_idx += 1
            else:
                z = layer(z, edge_index, edge_attr)
            if self.normalize != 'none':  # encoder.py:313-314
                z = self.norms[i](z)
            if i < self.num_layers - 1:  # encoder.py:315-317
                z = self._act(z)
                if self.training and self.p > 0 and dropout_masks is not None:
                    z = z * dropout_masks[i].to(z.dtype) / (1.0 - self.p)
                if getattr(self, "bf16_storage", False):
                    # emulation of the product's bf16 feature storage (no reference counterpart: the reference has no
                    # autocast call): a layer output that the next layer reads is rounded to bf16 (nearest-even) where
                    # the product stores it; arithmetic stays fp32 and the rounding is transparent to the gradient
                    z = z + (z.bfloat16().to(z.dtype) - z).detach()
        if reg_total is not None and self.training:
            self._last_env_reg = reg_total / env_idx
        else:
            self._last_env_reg = z.new_zeros(1)  # encoder.py:322
        return z

    def get_env_reg(self) -> Tensor:
        reg = self._last_env_reg if self._last_env_reg is not None else torch.zeros(1)
        self._last_env_reg = None
        return reg


class OracleInnerProductDecoder(nn.Module):
    """InnerProductDecoder (encoder.py:336-366) with hidden_dim given (proj_z=True)."""

    def __init__(self, hidden_dim: int, output_dim: int):
        super().__init__()
        self.lin = nn.Linear(hidden_dim, output_dim)

    def forward(self, z: Tensor, edge_index: Tensor, sigmoid: bool = True) -> Tensor:
        z = self.lin(z)
        value = (z[edge_index[0]] * z[edge_index[1]]).sum(dim=1)
        return torch.sigmoid(value) if sigmoid else value


# --------------------------------------------------------------------------------------
# a9-a11: VectorQuantize with CosineSimCodebook (model/vq.py:516-688, 692-1064), for the
# constructor arguments the entry scripts use: use_cosine_sim=True,
# separate_codebook_per_head=True, heads>=1, kmeans_init=False (pretrain.py:104-119).
# --------------------------------------------------------------------------------------
def l2norm(t: Tensor) -> Tensor:
    return F.normalize(t, p=2, dim=-1)  # vq.py:28-29 (eps 1e-12)


def orthogonal_loss(codes: Tensor) -> Tensor:
    """vq.py:232-237: sum(cos_sim^2) / (h * n^2) - 1/n over [h, n, d] codes."""
    h, n = codes.shape[:2]
    c = l2norm(codes)
    sim = torch.matmul(c, c.transpose(1, 2))
    return (sim ** 2).sum() / (h * n ** 2) - (1.0 / n)


class OracleCodebook(nn.Module):
    """Holds the CosineSimCodebook parameters/buffers under the reference's names
    (vq.py:563-571): embed [H,K,Dc] (Parameter when orthogonal loss is on, vq.py:785),
    initted [1], cluster_size [H,K], embed_avg [H,K,Dc]."""

    def __init__(self, heads: int, codebook_size: int, dim: int, learnable: bool):
        super().__init__()
        embed = torch.empty(heads, codebook_size, dim)
        nn.init.kaiming_uniform_(embed)  # uniform_init, vq.py:53-56
        embed = l2norm(embed)  # vq.py:541
        self.num_codebooks = heads
        self.register_buffer('initted', torch.Tensor([True]))
        self.register_buffer('cluster_size', torch.zeros(heads, codebook_size))
        self.register_buffer('embed_avg', embed.clone())
        if learnable:
            self.embed = nn.Parameter(embed)
        else:
            self.register_buffer('embed', embed)


class OracleVectorQuantize(nn.Module):
    def __init__(self, dim: int, codebook_size: int, codebook_dim: Optional[int] = None, heads: int = 1,
                 decay: float = 0.8, eps: float = 1e-5, commitment_weight: float = 1.0,
                 orthogonal_reg_weight: float = 0.0, orthogonal_reg_max_codes: Optional[int] = None,
                 ema_update: bool = True):
        super().__init__()
        codebook_dim = dim if codebook_dim is None else codebook_dim
        self.dim, self.heads, self.codebook_size, self.codebook_dim = dim, heads, codebook_size, codebook_dim
        inner = codebook_dim * heads
        self.has_projections = inner != dim  # vq.py:735-739
        self.project_in = nn.Linear(dim, inner) if self.has_projections else nn.Identity()
        self.project_out = nn.Linear(inner, dim) if self.has_projections else nn.Identity()
        self.decay, self.eps = decay, eps
        self.commitment_weight = commitment_weight
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self.orthogonal_reg_max_codes = orthogonal_reg_max_codes
        self.ema_update = ema_update
        self._codebook = OracleCodebook(heads, codebook_size, codebook_dim, learnable=orthogonal_reg_weight > 0)

    @property
    def codebook(self) -> Tensor:
        return self._codebook.embed

    def forward(self, z: Tensor, ortho_ids: Optional[Tensor] = None, tie_ind: Optional[Tensor] = None,
                tie_tol: float = 1e-5):
        """z [N, dim] -> (quantize [N, dim], embed_ind [N, H] int64, loss [1], orig_quantize [N, H*Dc]).

        ``ortho_ids`` replaces ``torch.randperm(K)[:max_codes]`` (vq.py:1024).
        ``tie_ind`` [N, H] (test aid for long replays): the assignment another implementation of this forward made.
        Where it differs from this arg-max but its similarity is within ``tie_tol`` of the maximum -- a near-tie, which
        fp32 summation order alone decides -- it is adopted, like any other replayed draw; a proposal further from the
        maximum is NOT adopted and counted in ``last_tie_rejected`` (the caller asserts zero).  ``last_tie_adopted``
        counts the adopted ones."""
        n = z.size(0)
        h, dc, k = self.heads, self.codebook_dim, self.codebook_size
        cb = self._codebook
        x = self.project_in(z)  # vq.py:881
        x = x.view(n, h, dc).permute(1, 0, 2)  # 'b n (h d) -> h b n d' with the n axis == 1 (vq.py:885-887)
        x = l2norm(x)  # vq.py:891, [H, N, Dc]
        embed = cb.embed
        # CosineSimCodebook.forward (vq.py:623-688), fp32 forced (vq.py:634)
        flat = x.float()
        sim = torch.einsum('hnd,hcd->hnc', flat, embed.detach() if not isinstance(embed, nn.Parameter) else embed)
        ind = sim.argmax(dim=-1)  # gumbel_sample with stochastic=False (vq.py:78-80): first max wins
        self.last_tie_adopted = self.last_tie_rejected = 0
        if tie_ind is not None:
            prop = tie_ind.reshape(n, h).permute(1, 0).to(ind.dtype)
            with torch.no_grad():
                differs = prop != ind
                short = sim.max(dim=-1).values - sim.gather(-1, prop.clamp(0, k - 1).unsqueeze(-1)).squeeze(-1)
                adopt = differs & (short <= tie_tol) & (prop >= 0) & (prop < k)
            self.last_tie_adopted = int(adopt.sum())
            self.last_tie_rejected = int((differs & ~adopt).sum())
            ind = torch.where(adopt, prop, ind)
        if self.training:
            onehot = F.one_hot(ind, k).to(flat.dtype)
            quant = torch.einsum('hnc,hcd->hnd', onehot, embed)  # vq.py:655-657
        else:
            quant = torch.gather(embed, 1, ind.unsqueeze(-1).expand(h, n, dc))  # vq.py:658-659
        if self.training and self.ema_update:  # vq.py:661-682
            with torch.no_grad():
                bins = onehot.sum(dim=1)
                cb.cluster_size.lerp_(bins, 1 - self.decay)
                embed_sum = torch.einsum('hnd,hnc->hcd', flat, onehot)
                cb.embed_avg.lerp_(embed_sum, 1 - self.decay)
                cs = cb.cluster_size
                smoothed = (cs + self.eps) / (cs.sum(dim=-1, keepdim=True) + k * self.eps)  # vq.py:102-104
                smoothed = smoothed * cs.sum(dim=-1, keepdim=True)
                cb.embed.data.copy_(l2norm(l2norm(cb.embed_avg / smoothed.unsqueeze(-1))))
        loss = torch.zeros(1)
        if self.training:
            commit_q = quant.detach()  # VectorQuantize.learnable_codebook is False (vq.py:747, 931-933)
            quant = x + (quant - x).detach()  # straight-through (vq.py:937)
            if self.commitment_weight > 0:
                loss = loss + F.mse_loss(commit_q, x) * self.commitment_weight  # vq.py:1007-1009
            if self.orthogonal_reg_weight > 0:  # vq.py:1011-1028
                codes = cb.embed
                if self.orthogonal_reg_max_codes is not None and k > self.orthogonal_reg_max_codes:
                    assert ortho_ids is not None, "inject the randperm ids"
                    codes = codes[:, ortho_ids]
                loss = loss + orthogonal_loss(codes) * self.orthogonal_reg_weight
        embed_ind = ind.permute(1, 0).contiguous()  # 'h b n -> b n h', n == 1 squeezed (vq.py:969-979)
        if h == 1:  # heads == 1 is not "multiheaded" (vq.py:865): no head axis on the indices
            embed_ind = embed_ind.view(n)
        orig_quantize = quant.permute(1, 0, 2).reshape(n, h * dc)  # 'h b n d -> b n (h d)' (vq.py:1034)
        out = self.project_out(orig_quantize)  # vq.py:1041
        return out, embed_ind, loss, orig_quantize


# --------------------------------------------------------------------------------------
# bf16 GEMM mode (BASELINE config 5 "bf16").  The reference has no autocast call; this
# emulates what the library's stemgnn_linear_set_mode(2) does to every nn.Linear on the path: BOTH operands of a
# product rounded to bf16 (nearest even), fp32 accumulation, fp32 bias; in the backward the incoming gradient is
# rounded the same way for the two products it enters, the bias gradient stays its fp32 column sum.  Everything that
# is not an nn.Linear (the quantiser's similarity einsum, vq.py:623,634; BatchNorm; the losses) stays fp32.
# --------------------------------------------------------------------------------------
class _Bf16Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xb, wb = x.detach().bfloat16().float(), w.detach().bfloat16().float()
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = b is not None
        y = xb @ wb.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, gy):
        xb, wb = ctx.saved_tensors
        gb = gy.bfloat16().float()
        g2, x2 = gb.reshape(-1, gb.shape[-1]), xb.reshape(-1, xb.shape[-1])
        return gb @ wb, g2.t() @ x2, (gy.reshape(-1, gy.shape[-1]).sum(0) if ctx.has_bias else None)


class bf16_gemms:
    """Context manager: every ``F.linear`` (hence every ``nn.Linear``) inside runs as ``_Bf16Linear``."""

    def __enter__(self):
        self._orig = torch.nn.functional.linear
        torch.nn.functional.linear = lambda input, weight, bias=None: _Bf16Linear.apply(input, weight, bias)
        return self

    def __exit__(self, *exc):
        torch.nn.functional.linear = self._orig
        return False


# --------------------------------------------------------------------------------------
# PyG 2.3.0 utilities used by pretrain.py:41-44 and pt_model.py:60, with the random
# draw passed in.  PARITY UNPINNED (PyG absent): restated from PyG 2.3.0 semantics.
# --------------------------------------------------------------------------------------
def mask_feature_col(x: Tensor, keep_cols: Tensor) -> Tensor:
    """torch_geometric.utils.mask_feature(x, p, mode='col'): one Bernoulli keep mask
    per feature column shared by all rows (``keep = rand(D) >= p``); masked -> 0."""
    return x.masked_fill(~keep_cols.view(1, -1), 0.0)


def dropout_adj_undirected(edge_index: Tensor, edge_attr: Optional[Tensor], keep: Tensor):
    """torch_geometric.utils.dropout_adj(..., force_undirected=True): ``keep =
    rand(E) >= p``; entries with row > col are dropped first; survivors are emitted
    in both directions ([row;col] then [col;row]) with edge_attr duplicated."""
    row, col = edge_index[0], edge_index[1]
    m = keep.clone()
    m[row > col] = False
    row, col = row[m], col[m]
    out_index = torch.stack([torch.cat([row, col]), torch.cat([col, row])], dim=0)
    out_attr = None
    if edge_attr is not None:
        ea = edge_attr[m]
        out_attr = torch.cat([ea, ea], dim=0)
    return out_index, out_attr, m


def check_negative_edges(neg_edge_index: Tensor, pos_edge_index: Tensor, num_nodes: int) -> bool:
    """Properties PyG's structured negative_sampling guarantees: in range, no
    self-loops, not a positive edge."""
    r, c = neg_edge_index[0], neg_edge_index[1]
    if r.numel() == 0:
        return True
    if int(r.min()) < 0 or int(c.min()) < 0 or int(r.max()) >= num_nodes or int(c.max()) >= num_nodes:
        return False
    if bool((r == c).any()):
        return False
    pos = set((pos_edge_index[0] * num_nodes + pos_edge_index[1]).tolist())
    return not any(v in pos for v in (r * num_nodes + c).tolist())


def cosine_lr_lambda(step: int, epochs: int) -> float:
    """utils/others.py:138-145 (stepped once per BATCH by pretrain.py:64-65)."""
    return (1 + math.cos(step * math.pi / epochs)) * 0.5


# --------------------------------------------------------------------------------------
# a7/a8: PretrainModel (model/pt_model.py:11-142)
# --------------------------------------------------------------------------------------
class OraclePretrainModel(nn.Module):
    def __init__(self, encoder: OracleEncoder, vq: OracleVectorQuantize, feat_recon_decoder: nn.Module,
                 topo_recon_decoder: OracleInnerProductDecoder, topo_sem_recon_decoder: nn.Module):
        super().__init__()
        self.encoder = encoder
        self.vq = vq
        self.feat_recon_decoder = feat_recon_decoder
        self.topo_recon_decoder = topo_recon_decoder
        self.topo_sem_recon_decoder = topo_sem_recon_decoder
        self.sem_encoder = deepcopy(encoder)  # pt_model.py:22
        self.sem_projector = nn.Linear(encoder.hidden_dim, encoder.hidden_dim)  # pt_model.py:23

    def forward(self, aug_g, g, bs: int, draws: Dict[str, Tensor]):
        """draws: 'student_dropout' / 'teacher_dropout' (lists of keep masks), 'topo_perm',
        'neg_edge_index', 'topo_sem_perm' (edge ids, already truncated to E_s), 'ortho_ids'."""
        x, ei, ea = aug_g
        ox, oei, oea = g
        z = self.encoder(x, ei, ea, dropout_masks=draws.get('student_dropout'))  # pt_model.py:112
        quantize, indices, commit_loss, _ = self.vq(z, ortho_ids=draws.get('ortho_ids'),
                                                    tie_ind=draws.get('vq_indices'),
                                                    tie_tol=float(draws.get('vq_tie_tol', 1e-5)))  # pt_model.py:113
        env_reg = self.encoder.get_env_reg()
        q = quantize
        # feat_recon_loss, pt_model.py:42-43
        feat = F.mse_loss(self.feat_recon_decoder(q[:bs]), ox[:bs])
        # topo_recon_loss, pt_model.py:46-65
        pos = oei[:, draws['topo_perm']]
        neg = draws['neg_edge_index']
        pos_loss = -torch.log(self.topo_recon_decoder(q, pos) + EPS).mean()
        neg_loss = -torch.log(1 - self.topo_recon_decoder(q, neg) + EPS).mean()
        topo = pos_loss + neg_loss
        # topo_sem_recon_loss, pt_model.py:68-83
        sel = draws['topo_sem_perm']
        e2 = oei[:, sel]
        cat = torch.cat([q[e2[0]], q[e2[1]]], dim=-1)
        topo_sem = F.mse_loss(self.topo_sem_recon_decoder(cat), oea[sel])
        # sem_recon_loss, pt_model.py:86-102 (teacher runs in train mode: BN batch stats + dropout)
        zt = self.sem_encoder(ox, oei, oea, dropout_masks=draws.get('teacher_dropout')).detach()
        hq = self.sem_projector(q)
        zt = F.normalize(zt[:bs], dim=-1, p=2)
        hq = F.normalize(hq[:bs], dim=-1, p=2)
        sem = (1 - (zt * hq).sum(dim=-1)).mean()
        losses = {'feat_recon_loss': feat, 'topo_recon_loss': topo, 'topo_sem_recon_loss': topo_sem,
                  'sem_recon_loss': sem, 'commit_loss': commit_loss, 'env_reg_loss': env_reg}
        return z, quantize, indices, losses

    @torch.no_grad()
    def ema_update_sem_encoder(self, decay: float = 0.99):  # pt_model.py:104-106 (parameters only)
        for pq, pk in zip(self.encoder.parameters(), self.sem_encoder.parameters()):
            pk.data = pk.data * decay + pq.data * (1 - decay)


def total_loss(losses: Dict[str, Tensor], params: Dict[str, float]) -> Tensor:
    """pretrain.py:51-58."""
    return (params['feat_lambda'] * losses['feat_recon_loss']
            + params['topo_lambda'] * losses['topo_recon_loss']
            + params['topo_sem_lambda'] * losses['topo_sem_recon_loss']
            + params['sem_lambda'] * losses['sem_recon_loss']
            + losses['commit_loss']
            + params.get('lamda_env', 0.0) * losses['env_reg_loss'])


def pretrain_step(model: OraclePretrainModel, optimizer, scheduler, params, x, edge_index, edge_attr, bs,
                  draws: Dict[str, Tensor]):
    """One iteration of pretrain.py:29-66 with the random draws injected.
    draws additionally holds 'feat_keep' [D] bool and 'edge_keep' [E] bool."""
    model.train()
    aug_x = mask_feature_col(x, draws['feat_keep'])  # pretrain.py:41
    aug_ei, aug_ea, _ = dropout_adj_undirected(edge_index, edge_attr, draws['edge_keep'])  # pretrain.py:42-44
    z, quantize, indices, losses = model((aug_x, aug_ei, aug_ea), (x, edge_index, edge_attr), bs, draws)
    # a replayed assignment ('vq_indices') may differ from this arg-max at near-ties only: anything else is a parity error
    assert model.vq.last_tie_rejected == 0, f"{model.vq.last_tie_rejected} code assignments differ beyond near-ties"
    loss = total_loss(losses, params)
    optimizer.zero_grad()
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), 1.0)  # pretrain.py:62
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    model.ema_update_sem_encoder(params['sem_encoder_decay'])  # pretrain.py:66
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, indices


def build_oracle_model(D: int, L: int, H: int, K: int, Dc: int, *, dropout=0.15, normalize='batch',
                       commit_weight=10.0, ortho_w=1.0, ortho_max=32, decay=0.8, ema_update=False,
                       negative_slope=0.0) -> OraclePretrainModel:
    """Model construction of pretrain.py:91-130."""
    enc = OracleEncoder(D, D, L, negative_slope=negative_slope, normalize=normalize, dropout=dropout)
    vq = OracleVectorQuantize(D, K, Dc, H, decay=decay, commitment_weight=commit_weight,
                              orthogonal_reg_weight=ortho_w, orthogonal_reg_max_codes=ortho_max,
                              ema_update=ema_update)
    return OraclePretrainModel(enc, vq, nn.Linear(D, D), OracleInnerProductDecoder(D, D), nn.Linear(2 * D, D))


# --------------------------------------------------------------------------------------
# Finetune consumer of the path (model/ft_model.py, task/node.py).  PARITY UNPINNED (no
# reference fixtures; restated from the source).
# --------------------------------------------------------------------------------------
def compute_multitask_loss(pred: Tensor, y: Tensor) -> Tensor:
    """ft_model.py:7-20: masked BCE-with-logits over the label matrix (0 -> -1 encodes 'negative',
    the in-place edit of ``y`` included)."""
    y[y == 0] = -1
    is_valid = y ** 2 > 0
    loss = 0.0
    for idx in range(y.shape[1]):
        exist_y = y[is_valid[:, idx], idx]
        exist_pred = pred[is_valid[:, idx], idx]
        loss = loss + F.binary_cross_entropy_with_logits(exist_pred.double(), (exist_y + 1) / 2, reduction="none").sum()
    return loss / torch.sum(is_valid)


class OracleTaskModel(nn.Module):
    """TaskModel (ft_model.py:23-107): linear decoder on top of encoder + VQ."""

    def __init__(self, encoder: OracleEncoder, vq: OracleVectorQuantize, num_classes: int, params: Dict):
        super().__init__()
        self.encoder, self.vq = encoder, vq
        num_heads, _, code_dim = vq.codebook.shape  # ft_model.py:32
        self.num_classes = num_classes
        self.num_heads = num_heads
        self.separate_decoder_for_each_head = params["separate_decoder_for_each_head"]
        self.decoder_jac_coeff = params.get("decoder_jac_coeff", 0.0)
        self.use_vq = params.get("use_vq", 1)
        if self.separate_decoder_for_each_head:
            self.decoder = nn.Linear(code_dim * num_heads, num_classes * num_heads)  # ft_model.py:41
        else:
            self.decoder = nn.Linear(code_dim, num_classes)

    def decoder_jacobian_penalty(self) -> Tensor:  # ft_model.py:45-50
        if self.decoder_jac_coeff <= 0:
            return torch.zeros(())
        return self.decoder_jac_coeff * self.decoder.weight.pow(2).sum()

    def encode(self, x, edge_index, edge_attr=None, dropout_masks=None):
        return self.encoder(x, edge_index, edge_attr, dropout_masks=dropout_masks)

    def encode_graph(self, x, edge_index, edge_attr=None, batch=None, pool="mean", dropout_masks=None):
        """ft_model.py:61-69 with global_mean_pool / global_add_pool restated as segment sums over ``batch``."""
        z = self.encoder(x, edge_index, edge_attr, dropout_masks=dropout_masks)
        n = int(batch.max()) + 1
        out = torch.zeros(n, z.size(1), dtype=z.dtype).index_add_(0, batch, z)
        if pool == "mean":
            out = out / torch.bincount(batch, minlength=n).clamp(min=1).to(z.dtype).unsqueeze(1)
        elif pool != "sum":
            raise NotImplementedError(pool)
        return out

    def get_lin_logits(self, z: Tensor, ortho_ids: Optional[Tensor] = None) -> Tensor:  # ft_model.py:90-103
        if self.use_vq:
            quantize, _, _, codes = self.vq(z, ortho_ids)
            if self.separate_decoder_for_each_head:
                return self.decoder(codes).reshape(-1, self.num_heads, self.num_classes)
            return self.decoder(quantize).reshape(-1, 1, self.num_classes)
        if self.separate_decoder_for_each_head:
            return self.decoder(self.vq.project_in(z)).reshape(-1, self.num_heads, self.num_classes)
        return self.decoder(z).reshape(-1, 1, self.num_classes)

    def compute_activation_loss(self, z, y, task="single", ortho_ids=None):  # ft_model.py:82-88
        logits = self.get_lin_logits(z, ortho_ids).mean(1)
        if task == "single":
            return F.cross_entropy(logits, y)
        if task == "multi":
            return compute_multitask_loss(logits, y)
        raise ValueError('task must be either "single" or "multi"')

    def forward(self, x, edge_index, edge_attr=None, dropout_masks=None):
        return self.get_lin_logits(self.encode(x, edge_index, edge_attr, dropout_masks))


def ft_node_full_batch_step(model: OracleTaskModel, optimizer, x, edge_index, edge_attr, y, train_mask, params,
                            dropout_masks=None, ortho_ids=None, scheduler=None) -> Dict[str, Tensor]:
    """ft_node with loader=None (task/node.py:38-63): encode the whole graph, cross-entropy on the train nodes,
    decoder penalty, env regulariser, one optimizer step."""
    model.train()
    z = model.encode(x, edge_index, edge_attr, dropout_masks)
    act_loss = model.compute_activation_loss(z[train_mask], y[train_mask], ortho_ids=ortho_ids) * 1.0
    jac_loss = model.decoder_jacobian_penalty()
    env_loss = params.get("lamda_env", 0.0) * model.encoder.get_env_reg()
    loss = act_loss + jac_loss + env_loss
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    if scheduler:
        scheduler.step()
    return {"act_loss": act_loss.detach(), "jac_loss": jac_loss.detach(), "env_loss": env_loss.detach(),
            "loss": loss.detach()}


def eval_node_full_batch(model: OracleTaskModel, x, edge_index, edge_attr, y, split: Dict[str, Tensor]):
    """eval_node with loader=None (task/node.py:106-135): softmax of the head-mean logits, accuracy * 100 per mask
    (utils/eval.py:11-29; torchmetrics multiclass Accuracy == fraction of arg-max hits)."""
    model.eval()
    with torch.no_grad():
        z = model.encode(x, edge_index, edge_attr)
        pred = model.get_lin_logits(z).mean(1).softmax(dim=-1)
        hit = (pred.argmax(dim=-1) == y).float()
        out = {k: hit[m].mean().item() * 100 for k, m in (("train", split["train"]), ("val", split["valid"]),
                                                           ("test", split["test"]))}
    return out, pred


def edge_embeddings(z: Tensor, edge_index: Tensor) -> Tensor:
    """task/link.py:7-8: the mean of the two endpoint embeddings."""
    return (z[edge_index[0]] + z[edge_index[1]]) / 2


def ft_link_full_batch_step(model: OracleTaskModel, optimizer, x, edge_index, edge_attr, y, train_mask, params,
                            dropout_masks=None, ortho_ids=None, scheduler=None) -> Dict[str, Tensor]:
    """ft_link with loader=None (task/link.py:19-48): encode the graph, classify the TRAIN edges from the mean of
    their endpoint embeddings (labels = one class per edge, e.g. the relation type), one optimizer step."""
    model.train()
    z = model.encode(x, edge_index, edge_attr, dropout_masks)
    env_loss = params.get("lamda_env", 0.0) * model.encoder.get_env_reg()
    edge_z = edge_embeddings(z, edge_index[:, train_mask])
    act_loss = model.compute_activation_loss(edge_z, y[train_mask], ortho_ids=ortho_ids) * 1.0
    jac_loss = model.decoder_jacobian_penalty()
    loss = act_loss + jac_loss + env_loss
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    if scheduler:
        scheduler.step()
    return {"act_loss": act_loss.detach(), "jac_loss": jac_loss.detach(), "env_loss": env_loss.detach(),
            "loss": loss.detach()}


def eval_link_full_batch(model: OracleTaskModel, x, edge_index, edge_attr, y, split: Dict[str, Tensor]):
    """eval_link with loader=None (task/link.py:98-108,125-140): per-edge class probabilities, accuracy * 100 per mask."""
    model.eval()
    with torch.no_grad():
        z = model.encode(x, edge_index, edge_attr)
        pred = model.get_lin_logits(edge_embeddings(z, edge_index)).mean(1).softmax(dim=-1)
        hit = (pred.argmax(dim=-1) == y).float()
        out = {k: hit[m].mean().item() * 100 for k, m in (("train", split["train"]), ("val", split["valid"]),
                                                           ("test", split["test"]))}
    return out, pred


def get_train_node_idx(ptr, weights, generator=None):
    """Weighted seed list of one pretraining epoch (reference dataset/process_datasets.py:186-198), restated: member i
    of the union graph (nodes ptr[i] .. ptr[i+1]) contributes every node int(w_i) times and a random
    int(frac(w_i) * n_i)-subset of its nodes once more."""
    total = torch.tensor([], dtype=torch.long)
    for i, (s, e) in enumerate(zip(ptr[:-1].tolist(), ptr[1:].tolist())):
        arr = torch.arange(s, e)
        whole, frac = int(weights[i]), weights[i] - int(weights[i])
        left = arr.repeat(whole)
        right = arr[torch.randperm(arr.size(0), generator=generator)[: int(frac * arr.size(0))]]
        total = torch.cat([total, left, right])
    return total


def merge_member_graphs(members):
    """preprocess_dataset_list + Batch.from_data_list (process_datasets.py:166-182) for members given as dicts with
    x, xe, edge_index, node_text_feat, edge_text_feat: the shifted ids and the member node offsets."""
    x_start = xe_start = n_start = 0
    xs, xes, eis, ptr = [], [], [], [0]
    for m in members:
        xs.append(m["x"] + x_start)
        xes.append(m["xe"] + xe_start)
        eis.append(m["edge_index"] + n_start)
        x_start += m["node_text_feat"].shape[0]
        xe_start += m["edge_text_feat"].shape[0]
        n_start += m["x"].shape[0]
        ptr.append(n_start)
    return torch.cat(xs), torch.cat(xes), torch.cat(eis, dim=1), torch.tensor(ptr)
