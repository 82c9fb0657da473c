"""Host side of the pretraining hot path, mirroring reference STEM-GNN/pretrain.py:25-167 with
the same parameter names (config/pretrain.yaml), running every device operation through the
gfx950 kernels.  Differences from the reference's host loop, all outside the arithmetic:

* features stay resident on the device; a batch is (node ids, edge_index, edge type ids) and the
  [N, D] / [E, D] expansions of pretrain.py:33-38 happen on the device (the edge one never:
  edge attributes stay a (type table, type id) pair);
* losses are not ``.item()``-ed per step (8 host syncs per step in pretrain.py:68-77); they are
  returned as device tensors and logged by the caller at its own cadence.
"""
from __future__ import annotations

import gc
from typing import Dict, Optional

import torch
import torch.nn as nn
from torch.optim import AdamW

from . import ops
from .graph import EdgeTypeAttr, as_graph
from .model.encoder import Encoder, InnerProductDecoder
from .model.pt_model import PretrainModel
from .model.vq import VectorQuantize
from .utils.graph_utils import mask_feature
from .utils.others import get_scheduler


def default_params() -> Dict:
    """reference config/pretrain.yaml:1-29 (+ argparse defaults utils/args.py:4-58)."""
    return dict(seed=42, input_dim=768, hidden_dim=768, num_layers=2, activation="relu", backbone="sage",
                normalize="batch", dropout=0.15, code_dim=768, codebook_size=128, codebook_head=4,
                codebook_decay=0.8, commit_weight=10, ortho_reg_weight=1, ortho_reg_max_codes=32,
                pretrain_epochs=50, pretrain_lr=1e-4, pretrain_weight_decay=1e-5, pretrain_batch_size=1024,
                feat_p=0.2, edge_p=0.2, topo_recon_ratio=0.1, feat_lambda=100, topo_lambda=0.01,
                topo_sem_lambda=100, sem_lambda=1, sem_encoder_decay=0.99, use_schedular=True, lamda_env=0.0,
                moe=False, moe_layers="none", moe_experts=3, moe_tau=1.0)


def build_model(params: Dict, device) -> PretrainModel:
    """Model construction of reference pretrain.py:91-130."""
    act = nn.ReLU if params["activation"] == "relu" else nn.LeakyReLU
    encoder = Encoder(input_dim=params["input_dim"], hidden_dim=params["hidden_dim"], activation=act,
                      num_layers=params["num_layers"], backbone=params["backbone"], normalize=params["normalize"],
                      dropout=params["dropout"], moe=params.get("moe", False),
                      num_experts=params.get("moe_experts", 3), tau=params.get("moe_tau", 1.0),
                      moe_layers=params.get("moe_layers", "none"))
    vq = VectorQuantize(dim=params["hidden_dim"], codebook_size=params["codebook_size"],
                        codebook_dim=params["code_dim"], heads=params["codebook_head"],
                        separate_codebook_per_head=True, decay=params["codebook_decay"],
                        commitment_weight=params["commit_weight"], use_cosine_sim=True,
                        orthogonal_reg_weight=params["ortho_reg_weight"],
                        orthogonal_reg_max_codes=params["ortho_reg_max_codes"],
                        orthogonal_reg_active_codes_only=False, kmeans_init=False, ema_update=False)
    model = PretrainModel(encoder=encoder, vq=vq,
                          feat_recon_decoder=nn.Linear(params["hidden_dim"], params["input_dim"]),
                          topo_recon_decoder=InnerProductDecoder(hidden_dim=params["hidden_dim"],
                                                                 output_dim=params["hidden_dim"]),
                          topo_sem_recon_decoder=nn.Linear(params["hidden_dim"] * 2, params["hidden_dim"]))
    return model.to(device)


def build_optimizer(model: nn.Module, params: Dict):
    """reference pretrain.py:134-136: AdamW over ALL parameters (sem_encoder's never get grads)."""
    # same update rule as the reference's AdamW, applied by one launch over a table of the tensors
    if next(model.parameters()).is_cuda:
        opt = ops.FusedAdamW(model.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    else:
        opt = AdamW(model.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    sched = get_scheduler(opt, params["use_schedular"], params["pretrain_epochs"])
    return opt, sched


def _trainable(model):
    """The parameters that can receive gradients, listed once per module (walking the module tree every step costs
    more host time than the clip itself).  Set ``model._stemgnn_trainable = None`` after adding or freezing
    parameters."""
    plist = getattr(model, "_stemgnn_trainable", None)
    if plist is None:
        plist = [p for p in model.parameters() if p.requires_grad]
        object.__setattr__(model, "_stemgnn_trainable", plist)
    return plist


_LOSS_KEYS = ("feat_recon_loss", "topo_recon_loss", "topo_sem_recon_loss", "sem_recon_loss", "commit_loss",
              "env_reg_loss")


def total_loss(losses: Dict[str, torch.Tensor], params: Dict) -> torch.Tensor:
    """reference pretrain.py:51-58: feat_lambda*feat + topo_lambda*topo + topo_sem_lambda*topo_sem +
    sem_lambda*sem + commit + lamda_env*env, as one weighted-sum launch (stack + mul + sum on the CPU)."""
    w = (float(params["feat_lambda"]), float(params["topo_lambda"]), float(params["topo_sem_lambda"]),
         float(params["sem_lambda"]), 1.0, float(params.get("lamda_env", 0.0)))
    terms = [losses[k] for k in _LOSS_KEYS]
    if terms[0].is_cuda and all(t.numel() == 1 and t.dtype == torch.float32 for t in terms):
        return ops.WeightedSumFn.apply(w, *terms)
    dev = terms[0].device
    key = (w, dev)
    wt = _WEIGHT_CACHE.get(key)
    if wt is None:
        wt = _WEIGHT_CACHE[key] = torch.tensor(w, dtype=torch.float32, device=dev)
    stacked = torch.stack([t.reshape(()).float() for t in terms])
    return (stacked * wt).sum().reshape(1)


_WEIGHT_CACHE: Dict = {}


_ONES = {}


def _ones_like_loss(loss):
    """The seed gradient of ``loss.backward()`` (a tensor of ones shaped like the loss), allocated once per device and
    shape instead of filled by a launch every step."""
    key = (loss.device, tuple(loss.shape), loss.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones_like(loss)
        ops.UNIT_GRADIENTS.add(t.data_ptr())  # lets the weighted loss sum hand out its weights without a launch
    return t


def pretrain_step(model: PretrainModel, optimizer, scheduler, params: Dict, x, edge_index, edge_attr, bs: int,
                  draws: Optional[Dict] = None, record_draws: bool = True, no_codebook: bool = False,
                  grad_sync=None, forward_fn=None):
    """One iteration of reference pretrain.py:41-66 on device-resident inputs.

    x [N, D] fp32; edge_index int64 [2, E]; edge_attr dense [E, D], EdgeTypeAttr or None.
    Returns (loss, losses dict, draws).  With ``record_draws`` the dropout keep masks are
    materialised into ``draws`` (test aid: the kernels themselves never store a mask).
    ``grad_sync`` (optional callable) runs between backward and clipping (explicit all-reduce);
    ``forward_fn`` replaces ``model.__call__`` (e.g. the DistributedDataParallel wrapper, whose
    hooks overlap the RCCL gradient all-reduce with backward)."""
    draws_in = draws or {}
    etype = edge_attr.etype if isinstance(edge_attr, EdgeTypeAttr) else None
    g = as_graph(edge_index, x.size(0), etype)  # a loader may hand the batch over as GraphStructure already
    if etype is not None:
        g.ensure_transpose()
    graph = [x, g, edge_attr]
    if "feat_keep" in draws_in or params["feat_p"] <= 0:
        aug_x, fmask = mask_feature(x, p=params["feat_p"], keep=draws_in.get("feat_keep"))  # pretrain.py:41
        fkey = None
    else:
        aug_x, fkey = ops.mask_columns(x, params["feat_p"])  # mask_feature(mode='col'), one kernel
        fmask = None
    # dropout_adj(..., force_undirected=True) (pretrain.py:42-44) straight from the CSR views: no COO
    # round trip, no re-sort, no host sync; the augmented slots address the ORIGINAL edge attributes
    g_aug = g.dropout_undirected(params["edge_p"], keep=draws_in.get("edge_keep"))
    aug_attr = EdgeTypeAttr(edge_attr.table, None) if isinstance(edge_attr, EdgeTypeAttr) else edge_attr
    aug_graph = [aug_x, g_aug, aug_attr]

    z, quantize, indices, losses = (forward_fn or model)(aug_graph, graph, params["topo_recon_ratio"], bs=bs,
                                                         no_codebook=no_codebook, draws=draws_in)
    loss = total_loss(losses, params)

    optimizer.zero_grad(set_to_none=True)
    loss.backward(gradient=_ones_like_loss(loss))  # a cached 1.0: autograd's own seed is a fill launch per step
    if grad_sync is not None:
        grad_sync()
    grads = [p.grad for p in _trainable(model) if p.grad is not None]
    if isinstance(optimizer, ops.FusedAdamW) and 0 < len(grads) <= ops._CLIP_MAX:
        # clip_grad_norm_(…, 1.0) + AdamW (pretrain.py:62-63): norm and factor in two launches, the factor applied
        # by the optimizer kernel while it reads the gradients (p.grad itself stays unclipped)
        optimizer.step(grad_coef=ops.grad_norm_coef(grads, 1.0)[1:])
    else:
        ops.clip_grad_norm_(_trainable(model), 1.0)  # pretrain.py:62
        optimizer.step()
    if scheduler:
        scheduler.step()
    model.ema_update_sem_encoder(decay=params["sem_encoder_decay"])  # pretrain.py:66

    out_draws = {}
    if record_draws:
        n, d = x.shape
        out_draws = dict(model.last_draws)
        if fkey is not None:
            out_draws["feat_keep"] = ops.dropout_keep_mask(d, params["feat_p"], *fkey, x.device)
        else:
            out_draws["feat_keep"] = (fmask.view(-1) if params["feat_p"] > 0
                                      else torch.ones(d, dtype=torch.bool, device=x.device))
        out_draws["edge_keep"] = (draws_in["edge_keep"] if "edge_keep" in draws_in else
                                  ops.dropout_keep_mask(g.num_edges, params["edge_p"], *g_aug.keep_key, x.device))
        p = params["dropout"]
        out_draws["student_dropout"] = [ops.dropout_keep_mask(n * d, p, s, o, x.device).view(n, d)
                                        for (s, o) in model.encoder.last_dropout_keys]
        out_draws["teacher_dropout"] = [ops.dropout_keep_mask(n * d, p, s, o, x.device).view(n, d)
                                        for (s, o) in model.sem_encoder.last_dropout_keys]
        if model.vq.last_ortho_ids is not None:
            out_draws["ortho_ids"] = model.vq.last_ortho_ids
        out_draws["vq_indices"] = indices.detach()  # how this run resolved near-ties of the arg-max (oracle: tie_ind)
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, out_draws


def batch_features(data, device) -> torch.Tensor:
    """The batch's node feature rows, by the reference's rule (pretrain.py:30-35): ``node_text_feat[data.x]`` when
    ``data.x`` and ``data.node_text_feat`` have different row counts (x = row ids into a shared table, what
    HipNeighborSampler emits), ``node_text_feat`` itself when they match (features already sliced per batch node,
    what PyG's NeighborLoader emits for a node-level attribute).  The lookup runs on the device and, with
    validation on, raises IndexError for an out-of-range id like the reference's indexing does."""
    ntf = data.node_text_feat.to(device)
    if data.x.size(0) != ntf.size(0):
        return ops.gather_rows(ntf, data.x.to(device).long().contiguous())
    return ntf


def pretrain(model, loader, optimizer, params, scheduler=None, no_codebook=False, log_fn=None, grad_sync=None):
    """reference pretrain.py:25-79.  ``loader`` yields batches with attributes
    ``batch_size``, ``x`` (node ids into ``node_text_feat``) or features, ``edge_index``, ``xe``,
    ``node_text_feat`` [*, D], ``edge_text_feat`` [T, D] — the NeighborLoader batch contract
    (reference dataset/process_datasets.py:92-108).  Feature tables should live on the device."""
    model.train()
    device = next(model.parameters()).device
    last = None
    # keep the cyclic collector off the hot loop (a generation-2 pass costs tens of ms when it fires in a
    # step); autograd graphs are reference-counted away each step, the epoch end collects the rest
    gc_was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        last = _pretrain_epoch(model, loader, optimizer, params, scheduler, no_codebook, log_fn, grad_sync, device)
    finally:
        if gc_was_enabled:
            gc.enable()
    return last


def _pretrain_epoch(model, loader, optimizer, params, scheduler, no_codebook, log_fn, grad_sync, device):
    last = None
    for data in loader:
        bs = data.batch_size
        x = batch_features(data, device)
        graph = getattr(data, "graph", None)  # the HIP sampler hands the batch's CSR over ready-made
        edge_index = graph if graph is not None else data.edge_index.to(device)
        edge_attr = EdgeTypeAttr(data.edge_text_feat.to(device), data.xe.to(device))  # edge_text_feat[xe], lazily
        loss, losses, _ = pretrain_step(model, optimizer, scheduler, params, x, edge_index, edge_attr, bs,
                                        record_draws=False, no_codebook=no_codebook, grad_sync=grad_sync)
        last = (loss, losses)
        if log_fn is not None:
            log_fn(loss, losses)
    return last
