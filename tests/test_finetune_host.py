"""CPU tests of the finetune consumer's host logic (no GPU): metrics against their definitions, pooling, the
multi-task loss against the oracle's restatement, checkpoint helpers (safe loader, freeze, mask <-> index)."""
import torch
import torch.nn as nn

from oracle import stem_oracle as O  # checker only


def test_accuracy_and_auc_definitions():
    from stem_gnn_amd.utils.eval import eval_acc, eval_auc, evaluate
    torch.manual_seed(0)
    pred, y = torch.rand(500, 4), torch.randint(0, 4, (500,))
    mask = torch.rand(500) < 0.5
    assert abs(eval_acc(pred, y, mask) - (pred[mask].argmax(1) == y[mask]).float().mean().item()) < 1e-7
    assert abs(evaluate(pred, y, None, {"task": "node"}) - 100 * (pred.argmax(1) == y).float().mean().item()) < 1e-5
    s = torch.randint(0, 20, (300, 3)).float()          # many ties
    t = (torch.rand(300, 3) < 0.4).float()
    t[::5, 2] = float("nan")                            # missing labels are skipped (utils/eval.py:41)
    t[:, 1] = 1.0                                       # a single-class column is skipped (utils/eval.py:40)
    exp = []
    for i in (0, 2):
        v = t[:, i] == t[:, i]
        p, q = s[v, i][t[v, i] == 1], s[v, i][t[v, i] == 0]
        exp.append(((p[:, None] > q[None, :]).double().mean() + 0.5 * (p[:, None] == q[None, :]).double().mean()).item())
    assert abs(eval_auc(s, t) - sum(exp) / 2) < 1e-9


def test_pooling_and_multitask_loss():
    from stem_gnn_amd.model.ft_model import _segment_pool, compute_multitask_loss
    torch.manual_seed(1)
    z = torch.randn(50, 6)
    batch = torch.sort(torch.randint(0, 7, (50,))).values
    batch[-1] = 6  # the pooled size is batch.max() + 1, as in PyG
    for how, ref in (("sum", lambda r: r.sum(0)), ("mean", lambda r: r.mean(0)), ("max", lambda r: r.max(0).values)):
        out = _segment_pool(z, batch, how)
        assert out.shape == (7, 6)
        for g in range(7):
            rows = z[batch == g]
            if rows.numel():
                torch.testing.assert_close(out[g], ref(rows))
            else:  # a graph id without nodes pools to a zero row (what PyG's scatter leaves there)
                assert float(out[g].abs().max()) == 0.0
        # batch=None (the reference's default for encode_graph, ft_model.py:62): one graph, one pooled row
        torch.testing.assert_close(_segment_pool(z, None, how), ref(z).unsqueeze(0))
    empty = torch.tensor([0, 0, 2, 2, 2])  # id 1 has no nodes
    for how in ("sum", "mean", "max"):
        out = _segment_pool(z[:5], empty, how)
        assert out.shape == (3, 6) and float(out[1].abs().max()) == 0.0
    pred = torch.randn(40, 5)
    y = torch.randint(0, 2, (40, 5)).float()
    a = compute_multitask_loss(pred, y.clone())
    b = O.compute_multitask_loss(pred, y.clone())
    torch.testing.assert_close(a, b)
    # definition: mean BCE-with-logits over all entries (none missing here)
    torch.testing.assert_close(a.double(), nn.functional.binary_cross_entropy_with_logits(pred.double(), y.double()))
    # missing labels (NaN) are left out of the sum and of the count, and 0 is relabelled -1 in place
    y2 = y.clone()
    y2[::3, 1] = float("nan")
    y_arg = y2.clone()
    a2 = compute_multitask_loss(pred, y_arg)
    b2 = O.compute_multitask_loss(pred, y2.clone())
    torch.testing.assert_close(a2, b2)
    assert bool(((y_arg == -1) == (y2 == 0)).all())


def test_checkpoint_helpers(tmp_path):
    from stem_gnn_amd.utils.others import load_params, freeze_params, mask2idx, idx2mask
    lin = nn.Linear(4, 3)
    path = tmp_path / "encoder_5.pt"
    torch.save(lin.state_dict(), path)
    other = load_params(nn.Linear(4, 3), str(path))
    assert all(torch.equal(a, b) for a, b in zip(lin.state_dict().values(), other.state_dict().values()))
    assert all(not p.requires_grad for p in freeze_params(other).parameters())
    m = torch.tensor([True, False, True, True, False])
    assert torch.equal(idx2mask(mask2idx(m), 5), m)


def test_oracle_task_model_known_answers():
    """Hand-checkable: identity-like decoder, one head, eval-mode quantiser -> logits are decoder(project_out(code))."""
    torch.manual_seed(0)
    enc = O.OracleEncoder(8, 8, 1)
    vq = O.OracleVectorQuantize(8, 5, 8, 1, commitment_weight=1.0, ema_update=False)
    params = {"separate_decoder_for_each_head": False, "use_vq": 1}
    tm = O.OracleTaskModel(enc, vq, 3, params).eval()
    z = torch.randn(6, 8)
    logits = tm.get_lin_logits(z)
    assert tuple(logits.shape) == (6, 1, 3)
    codes = vq.codebook[0]
    ind = (nn.functional.normalize(z, dim=-1) @ codes.t()).argmax(-1)
    torch.testing.assert_close(logits[:, 0], tm.decoder(codes[ind]))
    y = torch.tensor([0, 1, 2, 0, 1, 2])
    torch.testing.assert_close(tm.compute_activation_loss(z, y), nn.functional.cross_entropy(logits.mean(1), y))
    assert float(tm.decoder_jacobian_penalty()) == 0.0
    tm.decoder_jac_coeff = 0.5
    torch.testing.assert_close(tm.decoder_jacobian_penalty(), 0.5 * tm.decoder.weight.pow(2).sum())
