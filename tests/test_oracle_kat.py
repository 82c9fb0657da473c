"""Hand-derived known-answer tests for the parts of the CPU oracle that cannot be pinned against
the reference by import (torch_geometric is absent: SURVEY.md §8c).  PARITY UNPINNED at the PyG
boundary: these KATs restate PyG 2.3.0's documented semantics."""
import math

import torch

from oracle import stem_oracle as O


def test_mean_aggregate_path_graph():
    # 0 -> 1 -> 2 -> 3 plus 3 isolated from incoming edges at node 0
    x = torch.tensor([[1.0, -1.0], [2.0, 3.0], [-4.0, 0.5], [7.0, 7.0]])
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    out = O.sage_mean_aggregate(x, ei, None)
    exp = torch.tensor([[0.0, 0.0], [1.0, 0.0], [2.0, 3.0], [0.0, 0.5]])
    assert torch.equal(out, exp)


def test_mean_aggregate_duplicates_selfloop_and_edge_attr():
    x = torch.tensor([[1.0, -2.0], [2.0, 4.0], [-1.0, 3.0]])
    ei = torch.tensor([[0, 0, 2, 1], [1, 1, 1, 1]])  # duplicate edge 0->1 counted twice, self loop 1->1
    ea = torch.tensor([[0.0, 0.0], [-5.0, 5.0], [0.5, 0.5], [0.0, -10.0]])
    out = O.sage_mean_aggregate(x, ei, ea)
    msgs = torch.relu(torch.stack([x[0] + ea[0], x[0] + ea[1], x[2] + ea[2], x[1] + ea[3]]))
    assert torch.allclose(out[1], msgs.sum(0) / 4.0)
    assert torch.equal(out[0], torch.zeros(2)) and torch.equal(out[2], torch.zeros(2))


def test_sage_conv_isolated_node_is_bias_plus_root():
    torch.manual_seed(0)
    conv = O.OracleSAGEConv(3, 3)
    x = torch.randn(4, 3)
    ei = torch.tensor([[0], [1]])
    out = conv(x, ei, None)
    exp = conv.lin_l.bias + conv.lin_r(x[3])
    assert torch.allclose(out[3], exp, atol=1e-6)


def test_dropout_adj_force_undirected():
    ei = torch.tensor([[0, 1, 2, 3, 2, 1], [1, 0, 3, 2, 2, 3]])
    ea = torch.arange(12.0).view(6, 2)
    keep = torch.tensor([True, True, False, True, True, True])
    out_ei, out_ea, m = O.dropout_adj_undirected(ei, ea, keep)
    # row > col entries (1->0, 3->2) are dropped first; 2->3 dropped by the draw; survivors: 0->1, 2->2, 1->3
    assert m.tolist() == [True, False, False, False, True, True]
    assert out_ei.tolist() == [[0, 2, 1, 1, 2, 3], [1, 2, 3, 0, 2, 1]]
    assert out_ei.size(1) % 2 == 0
    assert torch.equal(out_ea, torch.cat([ea[m], ea[m]]))


def test_mask_feature_masks_whole_columns():
    x = torch.ones(5, 4)
    keep = torch.tensor([True, False, True, False])
    out = O.mask_feature_col(x, keep)
    assert torch.equal(out, torch.tensor([[1.0, 0.0, 1.0, 0.0]]).expand(5, 4))


def test_scheduler_values():
    assert O.cosine_lr_lambda(0, 50) == 1.0
    assert abs(O.cosine_lr_lambda(25, 50) - 0.5) < 1e-12
    assert abs(O.cosine_lr_lambda(50, 50)) < 1e-12
    assert abs(O.cosine_lr_lambda(100, 50) - 1.0) < 1e-12  # stepped per batch: period 100 steps


def test_orthogonal_loss_of_orthonormal_codes_is_zero():
    codes = torch.eye(4).unsqueeze(0)  # [1, 4, 4]
    assert abs(float(O.orthogonal_loss(codes))) < 1e-7
    same = torch.ones(1, 4, 4)
    assert abs(float(O.orthogonal_loss(same)) - (1 - 0.25)) < 1e-6


def test_mixture_layer_reversed_direction():
    torch.manual_seed(0)
    layer = O.OracleMixtureSageLayer(2, 2, 3)
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    ei = torch.tensor([[0, 0], [1, 2]])  # row 0 receives mean(x[1], x[2]) -- NOT the PyG flow direction
    out = layer(x, ei)
    agg0 = (x[1] + x[2]) / 2
    exp0 = torch.einsum('d,kdo->ko', torch.cat([agg0, x[0]]), layer.weights) + x[0]
    assert torch.allclose(out[0], exp0, atol=1e-5)


def test_pretrain_step_runs_and_decreases_nothing_weird():
    torch.manual_seed(0)
    D, N, E, bs = 16, 40, 120, 10
    om = O.build_oracle_model(D, 2, 2, 8, D, ortho_max=4)
    opt = torch.optim.AdamW(om.parameters(), lr=1e-3)
    x = torch.randn(N, D)
    ei = torch.randint(0, N, (2, E))
    ea = torch.randn(E, D)
    es = max(int(E * 0.1), 1)
    draws = {"feat_keep": torch.rand(D) >= 0.2, "edge_keep": torch.rand(E) >= 0.2,
             "student_dropout": [torch.rand(N, D) >= 0.15], "teacher_dropout": [torch.rand(N, D) >= 0.15],
             "topo_perm": torch.randperm(E)[:es], "neg_edge_index": torch.randint(0, N, (2, es)),
             "topo_sem_perm": torch.randperm(E)[:es], "ortho_ids": torch.randperm(8)[:4]}
    params = dict(feat_lambda=100, topo_lambda=0.01, topo_sem_lambda=100, sem_lambda=1, sem_encoder_decay=0.99)
    before = [p.detach().clone() for p in om.sem_encoder.parameters()]
    loss, losses, ind = O.pretrain_step(om, opt, None, params, x, ei, ea, bs, draws)
    assert math.isfinite(float(loss)) and tuple(ind.shape) == (N, 2)
    # teacher moved by exactly (1 - decay) towards the (updated) student
    for b, pk, pq in zip(before, om.sem_encoder.parameters(), om.encoder.parameters()):
        assert torch.allclose(pk, b * 0.99 + pq.detach() * 0.01, atol=1e-6)
