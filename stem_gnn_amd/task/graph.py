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


def _predict_graphs(model, loader, device):
    preds, labels = [], []
    for batch in loader:
        z = _encode(model, batch, device)
        preds.append(model.get_lin_logits(z).mean(1).detach())
        labels.append(batch.y.to(device).to(torch.float64))
    return torch.cat(preds, dim=0), torch.cat(labels, dim=0)


def _evaluate_loader(model, loader, device, params):
    if loader is None or len(loader) == 0:
        return float("nan")
    pred, y = _predict_graphs(model, loader, device)
    return evaluate(pred, y, None, params)


def eval_graph(model, dataset, loader, split, labels, params, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.eval()
    device = get_device_from_model(model)
    train_loader, val_loader, test_loader = loader
    with torch.no_grad():
        return {"train": _evaluate_loader(model, train_loader, device, params),
                "val": _evaluate_loader(model, val_loader, device, params),
                "test": _evaluate_loader(model, test_loader, device, params), "metric": task2metric[params["task"]]}
