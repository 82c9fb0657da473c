"""Graph-level (multi-task) finetune / evaluation steps (reference STEM-GNN/task/graph.py) on the HIP path: each
loader batch is a disjoint union of small graphs with per-node / per-edge text rows and a ``batch`` vector; the
encoder runs on this package's kernels, mean pooling and the masked multi-task BCE follow the reference."""
import torch

from ..utils.eval import evaluate, task2metric
from ..utils.others import get_device_from_model


def _encode(model, batch, device):
    return model.encode_graph(batch.node_text_feat.to(device), batch.edge_index.to(device), batch.edge_text_feat.to(device),
                              batch.batch.to(device), pool="mean")


def ft_graph(model, dataset, loader, optimizer, split, labels, params, scheduler=None, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.train()
    device = get_device_from_model(model)
    lamda_env = params.get("lamda_env", 0.0)
    vals, n = torch.zeros(4, dtype=torch.float64, device=device), 0
    for batch in loader:
        y = batch.y.to(device).to(torch.float64)
        z = _encode(model, batch, device)
        env_reg = model.get_env_reg()
        act_loss = model.compute_activation_loss(z, y, task="multi") * 1.0
        jac_loss = model.decoder_jacobian_penalty()
        env_loss = lamda_env * env_reg
        loss = act_loss + jac_loss + env_loss
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if scheduler:
            scheduler.step()
        vals = vals + torch.stack([act_loss.detach().reshape(()).double(), jac_loss.detach().reshape(()).double(),
                                   env_loss.detach().reshape(()).double(), loss.detach().reshape(()).double()])
        n += 1
    act, jac, env, tot = (vals / max(n, 1)).tolist()
    return {"act_loss": act, "jac_loss": jac, "env_loss": env, "loss": tot}


def _split_metric(model, loader, device, params):
    """The task metric over one split's loader: mean-pooled graph embeddings -> per-head class logits averaged over
    the heads (reference task/graph.py:57-77), scored against the fp64 labels; an absent or empty split is NaN."""
    if loader is None or len(loader) == 0:
        return float("nan")
    scored = [(model.get_lin_logits(_encode(model, batch, device)).mean(1).detach(),
               batch.y.to(device).to(torch.float64)) for batch in loader]
    logits, targets = (torch.cat(col, dim=0) for col in zip(*scored))
    return evaluate(logits, targets, None, params)


def eval_graph(model, dataset, loader, split, labels, params, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.eval()
    device = get_device_from_model(model)
    train_loader, val_loader, test_loader = loader
    with torch.no_grad():
        scores = {name: _split_metric(model, ld, device, params)
                  for name, ld in zip(("train", "val", "test"), (train_loader, val_loader, test_loader))}
    scores["metric"] = task2metric[params["task"]]
    return scores
