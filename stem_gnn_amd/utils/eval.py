"""Metrics of the finetune consumer (reference STEM-GNN/utils/eval.py).  torchmetrics / sklearn are
not needed for what they compute here: multiclass accuracy is the fraction of arg-max hits, ROC-AUC the
rank statistic of the positive scores."""
import torch

task2metric = {"node": "acc", "link": "acc", "graph": "auc"}


def eval_acc(y_pred, y_true, mask=None):
    """utils/eval.py:20-29 (torchmetrics Accuracy(task='multiclass'), micro average)."""
    if mask is not None:
        y_pred, y_true = y_pred[mask], y_true[mask]
    return (y_pred.argmax(dim=-1) == y_true).float().mean().item()


def _auc_1d(score, target):
    """Area under the ROC curve with tied scores given their mid-rank (sklearn.metrics.roc_auc_score)."""
    score, target = score.double().reshape(-1), target.reshape(-1).bool()
    order = torch.argsort(score)
    s = score[order]
    ranks = torch.arange(1, s.numel() + 1, dtype=torch.float64, device=s.device)
    uniq, inv, cnt = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = torch.cumsum(cnt, 0).double()
    mid = ends - (cnt.double() - 1) / 2
    r = torch.empty_like(ranks)
    r[order] = mid[inv]
    n_pos, n_neg = int(target.sum()), int((~target).sum())
    return ((r[target].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg)).item()


def eval_auc(y_pred, y_true):
    """utils/eval.py:32-48: mean ROC-AUC over the label columns that hold both classes (NaN labels skipped)."""
    y_pred, y_true = y_pred.detach(), y_true.detach()
    roc = []
    for i in range(y_true.shape[1]):
        col = y_true[:, i]
        if int((col == 1).sum()) > 0 and int((col == 0).sum()) > 0:
            valid = col == col
            roc.append(_auc_1d(y_pred[valid, i], col[valid] == 1))
    return sum(roc) / len(roc)


def evaluate(pred, y, mask=None, params=None):
    metric = task2metric[params["task"]]
    if metric == "acc":
        return eval_acc(pred, y, mask) * 100
    if metric == "auc":
        return eval_auc(pred, y) * 100
    raise ValueError(f"Metric {metric} is not supported.")
