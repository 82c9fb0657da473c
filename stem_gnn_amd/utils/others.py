"""Host helpers of the pretraining loop (reference STEM-GNN/utils/others.py)."""
import os
import random

import numpy as np
import torch

from .. import ops


def seed_everything(seed: int) -> None:
    """utils/others.py:73-81, plus the Philox key of the fused dropout kernels."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    # the device-side draws (dropout, edge dropout, edge / negative samples) must differ between data-parallel
    # ranks that are seeded alike: the rank goes into the high half of the Philox key
    ops.manual_seed((int(seed) & 0xFFFFFFFF) | (int(os.environ.get("RANK", "0")) << 32))


class CosineLambdaLR:
    """What the reference builds with ``LambdaLR(optimizer, lambda t: (1 + cos(t*pi/epochs)) / 2)``
    (utils/others.py:138-145): the same learning rates through the same ``step()`` protocol, without
    LambdaLR's per-step bookkeeping (a closed-form rate needs no chained state; ~0.15 ms of host time per
    step at 3 ms steps)."""

    def __init__(self, optimizer, epochs: int):
        self.optimizer, self.epochs = optimizer, epochs
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _factor(self, t: int) -> float:
        return float((1 + np.cos(t * np.pi / self.epochs)) * 0.5)

    def _apply(self) -> None:
        f = self._factor(self.last_epoch)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = base * f

    def step(self) -> None:
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "epochs": self.epochs}

    def load_state_dict(self, state) -> None:
        self.last_epoch, self.base_lrs, self.epochs = state["last_epoch"], list(state["base_lrs"]), state["epochs"]
        self._apply()


def get_scheduler(optimizer, use_scheduler=True, epochs=1000):
    """utils/others.py:138-145: lambda(t) = (1 + cos(t*pi/epochs)) / 2, stepped by the caller
    once per BATCH (reference pretrain.py:64-65)."""
    if not use_scheduler:
        return None
    return CosineLambdaLR(optimizer, epochs)


def get_device_from_model(model):
    return next(model.parameters()).device


def active_code(encoder, vq, data):
    """utils/others.py:152-157: the codes a graph activates and their share of the codebook."""
    z = encoder(data.x, data.edge_index, data.edge_attr)
    _, indices, _, _ = vq(z)
    return indices.unique(), indices.unique().numel() / (vq.codebook_size * vq.heads)


def load_params(model, path):
    """utils/others.py:160-171: load an ``encoder_{e}.pt`` / ``vq_{e}.pt`` state dict.  The reference runs one
    forward of the quantiser first so its k-means initialiser has fired before ``initted`` is overwritten; the
    module here takes the flag from the state dict directly.  Tensors only are unpickled."""
    state = torch.load(path, map_location=get_device_from_model(model), weights_only=True)
    model.load_state_dict(state)
    return model


def freeze_params(model):
    for param in model.parameters():
        param.requires_grad = False
    return model


def mask2idx(mask):
    return torch.where(mask == True)[0]  # noqa: E712  (reference spelling)


def idx2mask(idx, num_nodes):
    mask = torch.zeros(num_nodes, dtype=torch.bool)
    mask[idx] = 1
    return mask
