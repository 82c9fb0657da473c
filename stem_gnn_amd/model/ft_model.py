"""TaskModel: the finetune / evaluation consumer of the encoder + VQ path.

Interface contract kept from reference STEM-GNN/model/ft_model.py (it is what task/*.py, finetune.py and the
checkpoint helpers call): ``TaskModel(encoder, vq, num_classes, params)`` with the state-dict keys ``encoder.*``,
``vq.*``, ``decoder.weight`` / ``decoder.bias``; the attributes ``num_classes``, ``num_heads``,
``separate_decoder_for_each_head``, ``decoder_jac_coeff``, ``use_vq``; and the methods listed in ``__all__`` below
with the reference's argument names.  The bodies are this package's own: the encoder, the quantiser and the decoder
run on the HIP kernels, pooling is a segment reduction on the device, and the multi-task loss is one masked
reduction instead of a Python loop over tasks.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .. import ops

__all__ = ["TaskModel", "compute_multitask_loss"]


def compute_multitask_loss(pred: Tensor, y: Tensor) -> Tensor:
    """Masked multi-label BCE-with-logits (ft_model.py:7-20): labels are {0, 1} or NaN (missing); the mean runs
    over the labelled entries, in fp64.  Like the reference this relabels ``y`` IN PLACE (0 -> -1), which callers
    that reuse ``y`` observe."""
    y.masked_fill_(y == 0, -1)
    labelled = (y * y) > 0                      # False exactly for NaN (missing) entries
    target = torch.where(labelled, (y + 1) / 2, torch.zeros_like(y))  # NaNs must not reach the loss kernel
    # fp64 logits against targets in y's dtype, as the reference calls it (the result dtype is torch's choice there too)
    per_entry = F.binary_cross_entropy_with_logits(pred.double(), target, reduction="none")
    return (per_entry * labelled).sum() / labelled.sum()


def _segment_pool(z: Tensor, batch: Optional[Tensor], how: str) -> Tensor:
    """global_{mean,add,max}_pool (what ft_model.py:62-69 takes from torch_geometric): rows of ``z`` reduced per
    graph id in ``batch``; the number of graphs is ``batch.max() + 1`` as in PyG, a graph id without nodes pools to a
    zero row, and ``batch=None`` (the reference's default) means ONE graph: a single pooled row over all nodes."""
    if batch is None:
        if how == "max":
            return z.amax(dim=0, keepdim=True)
        return z.sum(dim=0, keepdim=True) if how == "sum" else z.mean(dim=0, keepdim=True)
    graphs = int(batch.max().item()) + 1
    where = batch.view(-1, 1).expand_as(z)
    acc = z.new_zeros(graphs, z.size(1))
    if how == "max":
        return acc.scatter_reduce_(0, where, z, reduce="amax", include_self=False)
    acc.scatter_add_(0, where, z)
    if how == "sum":
        return acc
    sizes = torch.bincount(batch, minlength=graphs).clamp_(min=1)
    return acc / sizes.to(z.dtype).unsqueeze(1)


def _linear_head(lin: nn.Linear, t: Tensor) -> Tensor:
    """``lin(t)`` on the HIP dense product.  Its tiles want out_features % 4 == 0; class heads (7 classes, 40
    classes x 3 heads ...) are zero-padded to the next multiple and the padding columns sliced off again."""
    extra = (-lin.out_features) % 4
    if extra == 0 or not t.is_cuda:
        return ops.linear(t, lin)
    weight = F.pad(lin.weight, (0, 0, 0, extra))
    bias = None if lin.bias is None else F.pad(lin.bias, (0, extra))
    flat, _ = ops.LinearFn.apply(t.reshape(-1, t.shape[-1]), weight, None, None, bias, False)
    return flat[:, :lin.out_features].reshape(*t.shape[:-1], lin.out_features)


class TaskModel(nn.Module):
    def __init__(self, encoder, vq, num_classes, params):
        super().__init__()
        self.encoder, self.vq = encoder, vq
        heads, _, code_dim = vq.codebook.shape          # [H, K, Dc] (vq.py:810-817)
        self.num_classes = num_classes
        self.num_heads = vq._codebook.num_codebooks
        self.separate_decoder_for_each_head = params["separate_decoder_for_each_head"]
        self.decoder_jac_coeff = params.get("decoder_jac_coeff", 0.0)
        self.use_vq = params.get("use_vq", 1)
        per_head = bool(self.separate_decoder_for_each_head)
        self.decoder = nn.Linear(code_dim * (heads if per_head else 1), num_classes * (heads if per_head else 1))

    # -- helpers -------------------------------------------------------------------------------------------------
    def _device(self):
        return next(self.parameters()).device

    @staticmethod
    def _get_linear_weight(module: nn.Module) -> Tensor:
        if not isinstance(module, nn.Linear):
            raise TypeError(f"Unsupported decoder module: {type(module).__name__}")
        return module.weight

    def _decoder_input(self, z: Tensor) -> Tuple[Tensor, int]:
        """What the decoder reads and how many logit groups it yields (ft_model.py:90-103): per-head decoders read
        the concatenated head vectors (quantised codes, or the raw ``project_in`` output without VQ), a shared
        decoder reads the D-wide vector (``project_out`` of the codes, or ``z`` itself)."""
        per_head = bool(self.separate_decoder_for_each_head)
        if self.use_vq:
            quantize, _, _, codes = self.vq(z)
            return (codes, self.num_heads) if per_head else (quantize, 1)
        if per_head:
            return self.vq._project(self.vq.project_in, z), self.num_heads
        return z, 1

    # -- the reference's method surface -----------------------------------------------------------------------------
    def decoder_jacobian_penalty(self) -> Tensor:
        """coeff * ||W_decoder||_F^2 (zero scalar when the coefficient is off)."""
        if self.decoder_jac_coeff > 0:
            return self.decoder_jac_coeff * self._get_linear_weight(self.decoder).square().sum()
        return torch.zeros((), device=self._device())

    def encode(self, x, edge_index, edge_attr=None):
        return self.encoder(x, edge_index, edge_attr)

    def encode_graph(self, x, edge_index, edge_attr=None, batch: Optional[Tensor] = None, pool="mean"):
        z = self.encode(x, edge_index, edge_attr)
        return _segment_pool(z, batch, pool) if pool in ("mean", "sum", "max") else z

    def get_env_reg(self, reset=True):
        fn = getattr(self.encoder, "get_env_reg", None)
        return fn(reset=reset) if fn is not None else torch.zeros(1, device=self._device())

    def get_moe_usage(self, reset=True):
        fn = getattr(self.encoder, "get_moe_usage", None)
        return fn(reset=reset) if fn is not None else []

    def get_lin_logits(self, z):
        """[N, heads or 1, num_classes] logits of the linear decoder."""
        source, groups = self._decoder_input(z)
        return _linear_head(self.decoder, source).reshape(-1, groups, self.num_classes)

    def compute_activation_loss(self, z, y, task="single"):
        if task not in ("single", "multi"):
            raise ValueError('task must be either "single" or "multi"')
        logits = self.get_lin_logits(z).mean(1)        # heads vote by averaging their logits
        return F.cross_entropy(logits, y) if task == "single" else compute_multitask_loss(logits, y)

    def forward(self, x, edge_index, edge_attr=None):
        return self.get_lin_logits(self.encode(x, edge_index, edge_attr))
