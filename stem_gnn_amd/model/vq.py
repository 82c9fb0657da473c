"""VectorQuantize with the reference's constructor / forward signature and state-dict keys
(reference STEM-GNN/model/vq.py:692-1064 + CosineSimCodebook :516-688), computing the
l2norm / similarity / arg-max / gather / straight-through / commitment chain in one fused
gfx950 kernel (csrc/vq.hip).

Scope: the configuration both entry scripts hard-code (pretrain.py:104-119, finetune.py:131-146):
``use_cosine_sim=True`` with one codebook per head.  The Euclidean codebook (vq.py:241-514) is
never instantiated by the reference and is not provided.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as distributed
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops


def l2norm(t: Tensor) -> Tensor:
    return F.normalize(t, p=2, dim=-1)  # vq.py:28-29


def orthogonal_loss_fn(t: Tensor) -> Tensor:
    """vq.py:232-237 (tiny: [H, <=max_codes, Dc]; plain torch ops)."""
    h, n = t.shape[:2]
    normed = l2norm(t)
    cosine_sim = torch.einsum("h i d, h j d -> h i j", normed, normed)
    return (cosine_sim ** 2).sum() / (h * n ** 2) - (1 / n)


def laplace_smoothing(x, n_categories, eps=1e-5, dim=-1):
    denom = x.sum(dim=dim, keepdim=True)
    return (x + eps) / (denom + n_categories * eps)  # vq.py:102-104


def _kmeans_cosine(samples: Tensor, num_clusters: int, num_iters: int):
    """vq.py:182-222 with use_cosine_sim=True on [H, n, d] samples (finetune-only init; the
    result is overwritten by load_state_dict right after, utils/others.py:167-170)."""
    h, n, d = samples.shape
    if n >= num_clusters:
        idx = torch.stack([torch.randperm(n, device=samples.device)[:num_clusters] for _ in range(h)])
    else:
        idx = torch.randint(0, n, (h, num_clusters), device=samples.device)
    means = torch.gather(samples, 1, idx.unsqueeze(-1).expand(h, num_clusters, d))
    bins = None
    for _ in range(num_iters):
        dists = samples @ means.transpose(1, 2)
        buckets = dists.argmax(dim=-1)
        bins = torch.zeros(h, num_clusters, dtype=buckets.dtype, device=samples.device)
        bins.scatter_add_(-1, buckets, torch.ones_like(buckets))
        zero_mask = bins == 0
        clamped = bins.masked_fill(zero_mask, 1)
        new_means = torch.zeros(h, num_clusters, d, dtype=samples.dtype, device=samples.device)
        new_means.scatter_add_(1, buckets.unsqueeze(-1).expand(h, n, d), samples)
        new_means = l2norm(new_means / clamped.unsqueeze(-1))
        means = torch.where(zero_mask.unsqueeze(-1), means, new_means)
    return means, bins


class CosineSimCodebook(nn.Module):
    """Parameter / buffer holder with the reference's names (vq.py:563-571)."""

    def __init__(self, dim, codebook_size, num_codebooks=1, kmeans_init=False, kmeans_iters=10, decay=0.8, eps=1e-5,
                 threshold_ema_dead_code=2, use_ddp=False, learnable_codebook=False, ema_update=True):
        super().__init__()
        self.decay, self.eps = decay, eps
        self.ema_update = ema_update
        self.codebook_size = codebook_size
        self.num_codebooks = num_codebooks
        self.kmeans_iters = kmeans_iters
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.use_ddp = use_ddp
        if not kmeans_init:
            embed = torch.empty(num_codebooks, codebook_size, dim)
            nn.init.kaiming_uniform_(embed)  # uniform_init, vq.py:53-56
            embed = l2norm(embed)
        else:
            embed = torch.zeros(num_codebooks, codebook_size, dim)
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self._initted_host = not kmeans_init  # host mirror of `initted`: no device sync per forward
        self.register_buffer("cluster_size", torch.zeros(num_codebooks, codebook_size))
        self.register_buffer("embed_avg", embed.clone())
        self.learnable_codebook = learnable_codebook
        if learnable_codebook:
            self.embed = nn.Parameter(embed)
        else:
            self.register_buffer("embed", embed)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        key = prefix + "initted"
        if key in state_dict:
            self._initted_host = bool(state_dict[key].reshape(-1)[0].item())

    def _all_reduce(self, t: Tensor) -> None:
        if self.use_ddp and distributed.is_initialized() and distributed.get_world_size() > 1:
            distributed.all_reduce(t)  # vq.py:666,672 (RCCL over xGMI on ROCm)

    @torch.no_grad()
    def init_embed_(self, xn: Tensor) -> None:
        """vq.py:573-595; xn [H, n, Dc] l2-normalised."""
        if self._initted_host:
            return
        embed, cluster_size = _kmeans_cosine(xn, self.codebook_size, self.kmeans_iters)
        self.embed.data.copy_(embed)
        self.embed_avg.data.copy_(embed * cluster_size.unsqueeze(-1))
        self.cluster_size.data.copy_(cluster_size)
        self.initted.data.fill_(1.0)
        self._initted_host = True

    @torch.no_grad()
    def ema_update_(self, xp: Tensor, ind: Tensor) -> None:
        """vq.py:661-682: EMA of per-code counts and sums, Laplace smoothing, renormalise."""
        h = self.num_codebooks
        norm = ops.vq_norms(xp, h).contiguous()
        bins, embed_sum = ops.vq_ema_stats(xp, norm, ind, self.codebook_size)
        self._all_reduce(bins)
        self.cluster_size.data.lerp_(bins, 1 - self.decay)
        self._all_reduce(embed_sum)
        self.embed_avg.data.lerp_(embed_sum, 1 - self.decay)
        cluster_size = laplace_smoothing(self.cluster_size, self.codebook_size, self.eps) * \
            self.cluster_size.sum(dim=-1, keepdim=True)
        embed_normalized = l2norm(self.embed_avg / cluster_size.unsqueeze(-1))
        self.embed.data.copy_(l2norm(embed_normalized))


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, heads=1, separate_codebook_per_head=False, decay=0.8,
                 eps=1e-5, freeze_codebook=False, kmeans_init=False, kmeans_iters=10, sync_kmeans=True,
                 use_cosine_sim=False, threshold_ema_dead_code=0, channel_last=True, accept_image_fmap=False,
                 commitment_weight=1.0, commitment_use_cross_entropy_loss=False, orthogonal_reg_weight=0.0,
                 orthogonal_reg_active_codes_only=False, orthogonal_reg_max_codes=None, stochastic_sample_codes=False,
                 sample_codebook_temp=1.0, straight_through=False, reinmax=False, sync_codebook=None,
                 sync_affine_param=False, ema_update=True, learnable_codebook=False, in_place_codebook_optimizer=None,
                 affine_param=False, affine_param_batch_decay=0.99, affine_param_codebook_decay=0.9,
                 sync_update_v=0.0):
        super().__init__()
        unsupported = {
            "use_cosine_sim=False (EuclideanCodebook)": not use_cosine_sim,
            "heads>1 without separate_codebook_per_head": heads > 1 and not separate_codebook_per_head,
            "accept_image_fmap": accept_image_fmap, "channel_last=False": not channel_last,
            "commitment_use_cross_entropy_loss": commitment_use_cross_entropy_loss,
            "stochastic_sample_codes": stochastic_sample_codes, "straight_through gumbel": straight_through,
            "reinmax": reinmax, "learnable_codebook": learnable_codebook,
            "in_place_codebook_optimizer": in_place_codebook_optimizer is not None, "affine_param": affine_param,
            "sync_update_v": sync_update_v > 0, "threshold_ema_dead_code>0": threshold_ema_dead_code > 0,
            "orthogonal_reg_active_codes_only": orthogonal_reg_active_codes_only,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("VectorQuantize options outside the reference's call sites "
                                      f"(pretrain.py:104-119, finetune.py:131-146): {bad}")
        self.dim = dim
        self.heads = heads
        self.separate_codebook_per_head = separate_codebook_per_head
        codebook_dim = dim if codebook_dim is None else codebook_dim
        codebook_input_dim = codebook_dim * heads
        requires_projection = codebook_input_dim != dim
        self.project_in = nn.Linear(dim, codebook_input_dim) if requires_projection else nn.Identity()
        self.project_out = nn.Linear(codebook_input_dim, dim) if requires_projection else nn.Identity()
        self.has_projections = requires_projection
        self.eps = eps
        self.commitment_weight = commitment_weight
        self.learnable_codebook = learnable_codebook
        self.has_codebook_orthogonal_loss = orthogonal_reg_weight > 0
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self.orthogonal_reg_max_codes = orthogonal_reg_max_codes
        self.codebook_dim = codebook_dim
        if sync_codebook is None:  # vq.py:771-772
            sync_codebook = distributed.is_initialized() and distributed.get_world_size() > 1
        self._codebook = CosineSimCodebook(
            dim=codebook_dim, num_codebooks=heads, codebook_size=codebook_size, kmeans_init=kmeans_init,
            kmeans_iters=kmeans_iters, decay=decay, eps=eps, threshold_ema_dead_code=threshold_ema_dead_code,
            use_ddp=sync_codebook, learnable_codebook=self.has_codebook_orthogonal_loss or learnable_codebook,
            ema_update=ema_update)
        self.codebook_size = codebook_size
        self.last_ortho_ids: Optional[Tensor] = None  # the randperm ids of the last training forward
        # Set by a caller that discards the fourth output (PretrainModel.quantize, pt_model.py:113): forward then
        # returns None in its place and never materialises the [N, H*Dc] per-head codes.
        self.skip_codes = False

    @property
    def codebook(self):
        codebook = self._codebook.embed
        return codebook if self.separate_codebook_per_head else codebook[0]

    def get_codes_from_indices(self, indices: Tensor) -> Tensor:
        """vq.py:827-843: indices [..., H] (or [...] for one head) -> codes [..., H*Dc]."""
        cb = self._codebook.embed
        if not self.separate_codebook_per_head:
            return cb[0][indices]
        lead = indices.shape[:-1]
        flat = indices.reshape(-1, self.heads)
        codes = torch.stack([cb[h][flat[:, h]] for h in range(self.heads)], dim=1)
        return codes.reshape(*lead, self.heads * cb.shape[-1])

    def get_output_from_indices(self, indices):
        return self.project_out(self.get_codes_from_indices(indices))

    @staticmethod
    def _rand_code_ids(num_codes, k, device):
        """torch.randperm(num_codes, device)[:k] (vq.py:1024): k distinct code ids."""
        if device.type == "cuda":
            return ops.sample_subset(num_codes, k, device)
        return torch.randperm(num_codes, device=device)[:k]

    @staticmethod
    def _project(lin, t):
        return ops.linear(t, lin) if isinstance(lin, nn.Linear) else lin(t)

    def _forward_phase(self, x, only_one, lead):
        """(quantize, embed_ind, loss, None) through ops.VqFn: the whole module as one library call per direction."""
        cb = self._codebook
        h = self.heads
        ids = None
        self.last_ortho_ids = None
        ortho = self.orthogonal_reg_weight if (self.training and self.has_codebook_orthogonal_loss) else 0.0
        if ortho:
            num_codes = cb.embed.shape[-2]
            if self.orthogonal_reg_max_codes is not None and num_codes > self.orthogonal_reg_max_codes:
                ids = self._rand_code_ids(num_codes, self.orthogonal_reg_max_codes, x.device)
                self.last_ortho_ids = ids
            else:
                ids = torch.arange(num_codes, device=x.device)
        cfg = dict(training=self.training, commit=self.commitment_weight if self.training else 0.0, ortho=ortho,
                   ortho_ids=ids)
        quantize, embed_ind, loss = ops.VqFn.apply(x, cb.embed, self.project_in.weight, self.project_in.bias,
                                                   self.project_out.weight, self.project_out.bias, cfg)
        if h == 1:
            embed_ind = embed_ind.view(x.size(0))
        if not only_one:
            quantize = quantize.reshape(*lead, -1)
            embed_ind = embed_ind.reshape(*lead, -1) if h > 1 else embed_ind.reshape(*lead)
        return quantize, embed_ind, loss, None

    def forward(self, x, indices=None, mask=None, sample_codebook_temp=None, freeze_codebook=False):
        if indices is not None or mask is not None:
            raise NotImplementedError("indices= / mask= are never passed by the reference's call sites")
        only_one = x.dim() == 2
        if not only_one:
            lead = x.shape[:-1]
            x = x.reshape(-1, x.shape[-1])
        n = x.size(0)
        h, dc = self.heads, self.codebook_dim
        cb = self._codebook
        # (the one-call phase reads project_out off a code table built in fp32: in the bf16 GEMM mode the module runs its
        # products as products, on rounded operands, through the per-op path)
        phase = (self.skip_codes and cb._initted_host and self.has_projections and x.is_cuda
                 and x.dtype == torch.float32 and not (self.training and cb.ema_update and not freeze_codebook)
                 and ops.linear_set_mode(-1) != 2)
        if phase:
            return self._forward_phase(x, only_one, None if only_one else lead)
        xp = self._project(self.project_in, x).float()  # vq.py:881; the codebook forces fp32 (vq.py:623,634)
        if not cb._initted_host:
            with torch.no_grad():
                cb.init_embed_(l2norm(xp.detach().view(n, h, dc).permute(1, 0, 2)))
        will_ema = self.training and cb.ema_update and not freeze_codebook
        commit = self.training and self.commitment_weight > 0
        quant, embed_ind, mse = ops.VqAssignFn.apply(xp, cb.embed, h, self.training, will_ema,
                                                     self.commitment_weight if commit else 1.0)
        if will_ema:
            cb.ema_update_(xp.detach().contiguous(), embed_ind)
        if self.training:
            terms = []
            if commit:
                terms.append(mse)  # commitment_weight * mse_loss(q.detach(), x), scaled inside the op (vq.py:1007-1009)
            if self.has_codebook_orthogonal_loss:  # vq.py:1011-1028
                codebook = cb.embed
                num_codes = codebook.shape[-2]
                if self.orthogonal_reg_max_codes is not None and num_codes > self.orthogonal_reg_max_codes:
                    rand_ids = self._rand_code_ids(num_codes, self.orthogonal_reg_max_codes, x.device)
                    self.last_ortho_ids = rand_ids
                else:
                    rand_ids = None
                    self.last_ortho_ids = None
                if codebook.is_cuda:
                    ids = rand_ids if rand_ids is not None else torch.arange(num_codes, device=x.device)
                    terms.append(ops.OrthoLossFn.apply(codebook, ids, self.orthogonal_reg_weight))  # fused fwd+bwd
                else:
                    sel = codebook if rand_ids is None else codebook[:, rand_ids]
                    terms.append(orthogonal_loss_fn(sel) * self.orthogonal_reg_weight)
            if terms:
                total = terms[0] if len(terms) == 1 else terms[0] + terms[1]
                loss = total.reshape(1)
            else:
                loss = torch.zeros(1, device=x.device, requires_grad=True)  # vq.py:983
        else:
            loss = torch.zeros(1, device=x.device)  # vq.py:983
        if h == 1:
            embed_ind = embed_ind.view(n)  # heads == 1 is not "multiheaded" (vq.py:865)
        orig_quantize = quant  # [N, H*Dc], heads already merged 'b n (h d)' (vq.py:1034)
        quantize = self._project(self.project_out, quant)  # vq.py:1041
        if not only_one:
            quantize = quantize.reshape(*lead, -1)
            orig_quantize = orig_quantize.reshape(*lead, -1)
            embed_ind = embed_ind.reshape(*lead, -1) if h > 1 else embed_ind.reshape(*lead)
        return quantize, embed_ind, loss, orig_quantize
