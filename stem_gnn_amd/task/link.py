"""Link-classification finetune / evaluation steps (reference STEM-GNN/task/link.py) on the HIP path: the encoder and
the quantiser run on this package's kernels; an edge is embedded as the mean of its endpoint embeddings
(task/link.py:7-8) and classified by ``TaskModel``.  Same arguments and returned dictionaries as the reference."""
import torch

from ..graph import EdgeTypeAttr
from ..utils.eval import evaluate, task2metric
from ..utils.others import get_device_from_model


def _edge_embeddings(z, edge_index):
    return (z[edge_index[0]] + z[edge_index[1]]) / 2


def _features(obj, device):
    ntf = obj.node_text_feat.to(device)
    x = getattr(obj, "x", None)
    if x is not None and x.dtype == torch.int64 and x.dim() == 1:
        return ntf[x.to(device)]  # node_text_feat[dataset.x] (task/link.py:20,101)
    return ntf


def _encode(model, obj, device):
    graph = getattr(obj, "graph", None)
    edge_index = graph if graph is not None else obj.edge_index.to(device)
    return model.encode(_features(obj, device), edge_index, EdgeTypeAttr(obj.edge_text_feat.to(device), obj.xe.to(device)))


def _step(model, optimizer, scheduler, edge_z, y, env_reg, lamda_env):
    act_loss = model.compute_activation_loss(edge_z, y) * 1.0
    jac_loss = model.decoder_jacobian_penalty()
    env_loss = lamda_env * env_reg
    loss = act_loss + jac_loss + env_loss
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    if scheduler:
        scheduler.step()
    return torch.stack([act_loss.detach().reshape(()), jac_loss.detach().reshape(()),
                        env_loss.detach().reshape(()), loss.detach().reshape(())])


def ft_link(model, dataset, loader, optimizer, split, labels, params, scheduler=None, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.train()
    device = get_device_from_model(model)
    lamda_env = params.get("lamda_env", 0.0)
    if loader is None:
        z = _encode(model, dataset, device)
        env_reg = model.get_env_reg()
        train_mask = split["train"].to(device)
        ei = dataset.edge_index.to(device)
        vals = _step(model, optimizer, scheduler, _edge_embeddings(z, ei[:, train_mask]), labels.to(device)[train_mask],
                     env_reg, lamda_env)
        n = 1
    else:
        vals, n = torch.zeros(4, device=device), 0
        for batch in loader:
            z = _encode(model, batch, device)
            env_reg = model.get_env_reg()
            edge_z = _edge_embeddings(z, batch.edge_label_index.to(device))
            vals = vals + _step(model, optimizer, scheduler, edge_z, batch.edge_label.to(device), env_reg, lamda_env)
            n += 1
    act, jac, env, tot = (vals / max(n, 1)).tolist()
    return {"act_loss": act, "jac_loss": jac, "env_loss": env, "loss": tot}


def eval_link(model, dataset, loader, split, labels, params, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.eval()
    device = get_device_from_model(model)
    with torch.no_grad():
        if loader is None:
            z = _encode(model, dataset, device)
            y = labels.to(device)
            pred = model.get_lin_logits(_edge_embeddings(z, dataset.edge_index.to(device))).mean(1).softmax(dim=-1)
        else:
            preds, gts = [], []
            for batch in loader:
                z = _encode(model, batch, device)
                edge_z = _edge_embeddings(z, batch.edge_label_index.to(device))
                preds.append(model.get_lin_logits(edge_z).mean(1).softmax(dim=-1).detach())
                gts.append(batch.edge_label.to(device))
            pred, y = torch.cat(preds, dim=0), torch.cat(gts, dim=0)
        masks = {k: split[k].to(pred.device) for k in ("train", "valid", "test")}
        return {"train": evaluate(pred, y, masks["train"], params), "val": evaluate(pred, y, masks["valid"], params),
                "test": evaluate(pred, y, masks["test"], params), "metric": task2metric[params["task"]]}
