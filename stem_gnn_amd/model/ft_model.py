"""TaskModel: the finetune / evaluation consumer of the encoder + VQ path (reference
STEM-GNN/model/ft_model.py).  Same constructor, attributes and methods; the encoder, the
quantiser and the linear decoder all run on the HIP kernels of this package."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


def compute_multitask_loss(pred, y):
    """ft_model.py:7-20 (the in-place 0 -> -1 relabelling of ``y`` included)."""
    criterion = nn.BCEWithLogitsLoss(reduction="none")
    y[y == 0] = -1
    is_valid = y ** 2 > 0
    loss = 0.0
    for idx in range(y.shape[1]):
        exist_y = y[is_valid[:, idx], idx]
        exist_pred = pred[is_valid[:, idx], idx]
        loss += torch.sum(criterion(exist_pred.double(), (exist_y + 1) / 2))
    return loss / torch.sum(is_valid)


def _pool(z, batch, how: str, size=None):
    """global_{mean,add,max}_pool of torch_geometric (ft_model.py:4,62-69): segment reduce of node rows by graph id."""
    n = int(batch.max().item()) + 1 if size is None else size
    idx = batch.view(-1, 1).expand_as(z)
    out = torch.zeros(n, z.size(1), dtype=z.dtype, device=z.device)
    if how == "sum":
        return out.scatter_add_(0, idx, z)
    if how == "mean":
        cnt = torch.bincount(batch, minlength=n).clamp(min=1).to(z.dtype).unsqueeze(1)
        return out.scatter_add_(0, idx, z) / cnt
    return out.scatter_reduce_(0, idx, z, reduce="amax", include_self=False)


def _decode(lin: nn.Linear, t):
    """nn.Linear through the HIP product.  The kernel wants out_features % 4 == 0; the class heads
    (e.g. 7 classes) are padded with zero rows and the padding columns dropped."""
    pad = (-lin.out_features) % 4
    if pad == 0 or not t.is_cuda:
        return ops.linear(t, lin)
    w = torch.cat([lin.weight, lin.weight.new_zeros(pad, lin.in_features)], dim=0)
    b = None if lin.bias is None else torch.cat([lin.bias, lin.bias.new_zeros(pad)])
    y, _ = ops.LinearFn.apply(t.reshape(-1, t.shape[-1]), w, None, None, b, False)
    return y[:, :lin.out_features].reshape(*t.shape[:-1], lin.out_features)


class TaskModel(nn.Module):
    """Linear decoder built on top of the encoder + VQ backbone (ft_model.py:23-107)."""

    def __init__(self, encoder, vq, num_classes, params):
        super().__init__()
        self.encoder = encoder
        self.vq = vq
        num_heads, _, code_dim = vq.codebook.shape
        self.num_classes = num_classes
        self.num_heads = vq._codebook.num_codebooks if vq is not None else 1
        self.separate_decoder_for_each_head = params["separate_decoder_for_each_head"]
        self.decoder_jac_coeff = params.get("decoder_jac_coeff", 0.0)
        self.use_vq = params.get("use_vq", 1)
        if self.separate_decoder_for_each_head:
            self.decoder = nn.Linear(code_dim * num_heads, num_classes * num_heads)
        else:
            self.decoder = nn.Linear(code_dim, num_classes)

    def decoder_jacobian_penalty(self):
        if self.decoder_jac_coeff <= 0:
            return torch.zeros((), device=next(self.parameters()).device)
        return self.decoder_jac_coeff * self._get_linear_weight(self.decoder).pow(2).sum()

    @staticmethod
    def _get_linear_weight(module: nn.Module):
        if isinstance(module, nn.Linear):
            return module.weight
        raise TypeError(f"Unsupported decoder module: {type(module).__name__}")

    def encode(self, x, edge_index, edge_attr=None):
        return self.encoder(x, edge_index, edge_attr)

    def encode_graph(self, x, edge_index, edge_attr=None, batch=None, pool="mean"):
        z = self.encoder(x, edge_index, edge_attr)
        if pool in ("mean", "sum", "max"):
            z = _pool(z, batch, pool)
        return z

    def get_env_reg(self, reset=True):
        if hasattr(self.encoder, "get_env_reg"):
            return self.encoder.get_env_reg(reset=reset)
        return torch.zeros(1, device=next(self.parameters()).device)

    def get_moe_usage(self, reset=True):
        if hasattr(self.encoder, "get_moe_usage"):
            return self.encoder.get_moe_usage(reset=reset)
        return []

    def compute_activation_loss(self, z, y, task="single"):
        logits = self.get_lin_logits(z).mean(1)
        if task == "single":
            return F.cross_entropy(logits, y)
        if task == "multi":
            return compute_multitask_loss(logits, y)
        raise ValueError('task must be either "single" or "multi"')

    def get_lin_logits(self, z):
        if self.use_vq:
            quantize, _, _, codes = self.vq(z)
            if self.separate_decoder_for_each_head:
                return _decode(self.decoder, codes).reshape(-1, self.num_heads, self.num_classes)
            return _decode(self.decoder, quantize).reshape(-1, 1, self.num_classes)
        if self.separate_decoder_for_each_head:
            codes = self.vq._project(self.vq.project_in, z)
            return _decode(self.decoder, codes).reshape(-1, self.num_heads, self.num_classes)
        return _decode(self.decoder, z).reshape(-1, 1, self.num_classes)

    def forward(self, x, edge_index, edge_attr=None):
        z = self.encoder(x, edge_index, edge_attr)
        return self.get_lin_logits(z)
