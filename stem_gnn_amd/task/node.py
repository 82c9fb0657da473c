"""Node-classification finetune / evaluation steps (reference STEM-GNN/task/node.py), on the HIP path.

Same arguments and returned dictionaries as the reference.  Batch contract (task/node.py:22-27 there): a batch is
moved to the model's device when it has ``.to`` (PyG batches do; the HIP sampler's are device-resident already),
and its node features follow ``pretrain.batch_features``: ``node_text_feat`` itself when it has one row per batch
node, ``node_text_feat[batch.x]`` when ``x`` holds row ids into a shared table.  Inside, the edge attribute of a
batch is handed over as (type table, type ids) instead of the gathered [E, D] rows (same numbers, half the
aggregation's traffic), and the four ``.item()`` syncs of a step are one."""
import torch

from ..graph import EdgeTypeAttr
from ..pretrain import batch_features
from ..utils.eval import evaluate, task2metric
from ..utils.others import get_device_from_model


def _edge_attr(obj, device):
    return EdgeTypeAttr(obj.edge_text_feat.to(device), obj.xe.to(device))


def _on_device(batch, device):
    """``batch.to(device)`` of the reference loop (task/node.py:22,71) for batches that can move themselves."""
    to = getattr(batch, "to", None)
    return to(device) if callable(to) else batch


def _run_full_batch(model, dataset, labels, split, params):
    device = get_device_from_model(model)
    x = dataset.node_text_feat.to(device)
    edge_index = getattr(dataset, "graph", None) or dataset.edge_index.to(device)
    z = model.encode(x, edge_index, _edge_attr(dataset, device))
    return z, labels.to(device)


def _accumulate_minibatch_predictions(model, loader, device):
    preds, gts = [], []
    for batch in loader:
        batch = _on_device(batch, device)
        bs = batch.batch_size
        x = batch_features(batch, device)
        graph = getattr(batch, "graph", None) or batch.edge_index
        z = model.encode(x, graph, _edge_attr(batch, device))[:bs]
        pred = model.get_lin_logits(z).mean(1).softmax(dim=-1)
        preds.append(pred.detach())
        gts.append(batch.y[:bs].to(device))
    return torch.cat(preds, dim=0), torch.cat(gts, dim=0)


def _step(model, optimizer, scheduler, z, y, lamda_env):
    act_loss = model.compute_activation_loss(z, y) * 1.0
    jac_loss = model.decoder_jacobian_penalty()
    env_loss = lamda_env * model.get_env_reg()
    loss = act_loss + jac_loss + env_loss
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    if scheduler:
        scheduler.step()
    return torch.stack([act_loss.detach().reshape(()), jac_loss.detach().reshape(()),
                        env_loss.detach().reshape(()), loss.detach().reshape(())])


def ft_node(model, dataset, loader, optimizer, split, labels, params, scheduler=None, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.train()
    device = get_device_from_model(model)
    lamda_env = params.get("lamda_env", 0.0)
    if loader is None:
        z, y = _run_full_batch(model, dataset, labels, split, params)
        train_mask = split["train"].to(z.device)
        vals = _step(model, optimizer, scheduler, z[train_mask], y[train_mask], lamda_env)
        n = 1
    else:
        vals, n = torch.zeros(4, device=device), 0
        for batch in loader:
            batch = _on_device(batch, device)
            bs = batch.batch_size
            x = batch_features(batch, device)
            graph = getattr(batch, "graph", None) or batch.edge_index
            z = model.encode(x, graph, _edge_attr(batch, device))[:bs]
            vals = vals + _step(model, optimizer, scheduler, z, batch.y[:bs].to(device), lamda_env)
            n += 1
    act, jac, env, tot = (vals / max(n, 1)).tolist()
    return {"act_loss": act, "jac_loss": jac, "env_loss": env, "loss": tot}


def eval_node(model, dataset, loader, split, labels, params, **kwargs):
    assert params["setting"] == "standard", "Only standard setting is supported"
    model.eval()
    device = get_device_from_model(model)
    with torch.no_grad():
        if loader is None:
            z, y = _run_full_batch(model, dataset, labels, split, params)
            pred = model.get_lin_logits(z).mean(1).softmax(dim=-1)
        else:
            pred, y = _accumulate_minibatch_predictions(model, loader, device)
        masks = {k: split[k].to(pred.device) for k in ("train", "valid", "test")}
        return {"train": evaluate(pred, y, masks["train"], params), "val": evaluate(pred, y, masks["valid"], params),
                "test": evaluate(pred, y, masks["test"], params), "metric": task2metric[params["task"]]}
