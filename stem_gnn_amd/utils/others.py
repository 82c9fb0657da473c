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
    ops.manual_seed(seed)


def get_scheduler(optimizer, use_scheduler=True, epochs=1000):
    """utils/others.py:138-145: lambda(t) = (1 + cos(t*pi/epochs)) / 2, stepped by the caller
    once per BATCH (reference pretrain.py:64-65)."""
    if not use_scheduler:
        return None
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda t: (1 + np.cos(t * np.pi / epochs)) * 0.5)


def get_device_from_model(model):
    return next(model.parameters()).device
