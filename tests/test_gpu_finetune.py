"""GPU parity tests of the finetune / evaluation consumer (BASELINE config 1: Cora-sized node classification,
full batch): TaskModel + ft_node / eval_node on the HIP path against the CPU oracle replaying the HIP run's dropout
draws.  No Cora data can be materialised here: a synthetic graph with Cora's node / edge / feature /
class counts and unit-norm features stands in."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import stem_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class Data:
    pass


def cora_like(seed=0, n=2708, e=10556, d=768, c=7, t=1):
    g = torch.Generator().manual_seed(seed)
    half = torch.randint(0, n, (2, e // 2), generator=g)
    ei = torch.cat([half, half.flip(0)], dim=1)
    data = Data()
    data.node_text_feat = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    data.edge_text_feat = torch.nn.functional.normalize(torch.randn(t, d, generator=g), dim=-1)
    data.xe = torch.randint(0, t, (ei.size(1),), generator=g)
    data.edge_index = ei
    labels = torch.randint(0, c, (n,), generator=g)
    perm = torch.randperm(n, generator=g)
    split = {}
    for name, lo, hi in (("train", 0, 140), ("valid", 140, 640), ("test", 640, 1640)):
        m = torch.zeros(n, dtype=torch.bool)
        m[perm[lo:hi]] = True
        split[name] = m
    return data, labels, split


def build_pair(D, L, H, K, C, dev, params, dropout=0.15, normalize="none", seed=0):
    from stem_gnn_amd.model.encoder import Encoder
    from stem_gnn_amd.model.ft_model import TaskModel
    from stem_gnn_amd.model.vq import VectorQuantize
    torch.manual_seed(seed)
    oenc = O.OracleEncoder(D, D, L, normalize=normalize, dropout=dropout)
    ovq = O.OracleVectorQuantize(D, K, D, H, decay=0.8, commitment_weight=10, orthogonal_reg_weight=1,
                                 orthogonal_reg_max_codes=32, ema_update=False)
    om = O.OracleTaskModel(oenc, ovq, C, params)
    enc = Encoder(D, D, nn.ReLU, L, backbone="sage", normalize=normalize, dropout=dropout)
    vq = VectorQuantize(dim=D, codebook_size=K, codebook_dim=D, heads=H, separate_codebook_per_head=True, decay=0.8,
                        commitment_weight=10, use_cosine_sim=True, orthogonal_reg_weight=1,
                        orthogonal_reg_max_codes=32, kmeans_init=False, ema_update=False)
    gm = TaskModel(enc, vq, C, params)
    gm.load_state_dict(om.state_dict())  # the reference's key names on both sides
    return om, gm.to(dev)


@pytest.mark.parametrize("separate,use_vq,freeze", [(True, 1, 1), (False, 1, 0), (True, 0, 1)])
def test_cora_sized_finetune_steps_and_eval(dev, separate, use_vq, freeze):
    from stem_gnn_amd import ops
    from stem_gnn_amd.task.node import ft_node, eval_node
    from stem_gnn_amd.utils.others import freeze_params
    D, L, H, K, C = 768, 2, 4, 128, 7
    params = {"separate_decoder_for_each_head": separate, "decoder_jac_coeff": 1e-3, "use_vq": use_vq,
              "setting": "standard", "task": "node", "lamda_env": 0.0}
    data, labels, split = cora_like()
    om, gm = build_pair(D, L, H, K, C, dev, params)
    if freeze:  # finetune.py:178-180
        freeze_params(om.vq); freeze_params(gm.vq)
    opt_o = torch.optim.AdamW([p for p in om.parameters()], lr=5e-4)   # config/finetune.yaml node.cora
    opt_g = torch.optim.AdamW([p for p in gm.parameters()], lr=5e-4)
    ops.manual_seed(3)
    N = data.node_text_feat.size(0)
    ea_cpu = data.edge_text_feat[data.xe]
    for step in range(3):
        # the quantiser's orthogonal-loss draw is unused downstream (its loss output is discarded by TaskModel)
        ids = torch.arange(32)
        gm.vq._rand_code_ids = lambda n, k, device: ids.to(device)
        out_g = ft_node(gm, data, None, opt_g, split, labels, params)
        masks = [ops.dropout_keep_mask(N * D, 0.15, s, o, dev).view(N, D).cpu() for (s, o) in gm.encoder.last_dropout_keys]
        out_o = O.ft_node_full_batch_step(om, opt_o, data.node_text_feat, data.edge_index, ea_cpu, labels,
                                          split["train"], params, dropout_masks=masks, ortho_ids=ids)
        for k in ("act_loss", "jac_loss", "env_loss", "loss"):
            assert abs(out_g[k] - float(out_o[k])) <= 1e-4 * max(1.0, abs(float(out_o[k]))), (step, k, out_g, out_o)
    # parameters after three AdamW steps (lr 5e-4: Adam normalises the gradient, so this bounds its direction too)
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        assert n1 == n2
        if p1.requires_grad:
            torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-3, atol=2e-4, msg=lambda m: f"{n1}: {m}")
    res_g = eval_node(gm, data, None, split, labels, params)
    res_o, pred_o = O.eval_node_full_batch(om, data.node_text_feat, data.edge_index, ea_cpu, labels, split)
    assert res_g["metric"] == "acc"
    for k in ("train", "val", "test"):
        # accuracy moves in steps of 100/|mask|: allow one near-tie arg-max flip per mask
        assert abs(res_g[k] - res_o[k]) <= 100.0 / int(split["valid" if k == "val" else k].sum()) + 1e-6
    gm.eval()
    with torch.no_grad():
        logits = gm(data.node_text_feat.to(dev), data.edge_index.to(dev), ea_cpu.to(dev)).mean(1).softmax(-1)
    torch.testing.assert_close(logits.cpu(), pred_o, rtol=1e-4, atol=1e-5)


def test_minibatch_finetune_and_eval_run_on_the_hip_loader(dev):
    """ft_node / eval_node with a loader (finetune.py:198-200: NeighborLoader batches), on the HIP sampler: every
    labelled seed is predicted exactly once and the loss goes down."""
    from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader
    from stem_gnn_amd.task.node import ft_node, eval_node
    D, L, H, K, C = 128, 2, 4, 64, 7
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 0.0, "use_vq": 1, "setting": "standard",
              "task": "node"}
    data, labels, split = cora_like(seed=1, d=D, t=3)
    _, gm = build_pair(D, L, H, K, C, dev, params, dropout=0.0)
    n = data.node_text_feat.size(0)
    ei, xe = data.edge_index.to(dev), data.xe.to(dev)
    sampler = HipNeighborSampler(ei, xe, n, torch.arange(n, device=dev), data.node_text_feat.to(dev),
                                 data.edge_text_feat.to(dev), [30, 30], seed=0)
    y_dev = labels.to(dev)

    class Loader:
        def __init__(self, nodes, shuffle):
            self.inner = NeighborLoader(sampler, nodes, 512, shuffle=shuffle)

        def __iter__(self):
            for b in self.inner:
                b.y = y_dev[b.n_id]
                yield b

        def __len__(self):
            return len(self.inner)

    train_nodes = torch.where(split["train"])[0].to(dev)
    opt = torch.optim.AdamW(gm.parameters(), lr=5e-3)
    losses = [ft_node(gm, data, Loader(train_nodes, True), opt, split, labels, params)["loss"] for _ in range(8)]
    assert losses[-1] < losses[0]
    res = eval_node(gm, data, Loader(torch.arange(n, device=dev), False), split, labels, params)
    assert 0.0 <= res["test"] <= 100.0 and res["train"] > 100.0 / C  # better than chance on the nodes it trained on


def test_get_loader_node_task_full_neighbourhood_eval_equals_full_batch(dev):
    """utils/loader.get_loader (reference utils/loader.py:9-26): [10] * L training loader, [-1] * L evaluation loader in
    batches of 512.  With every in-neighbour of every hop in the batch, a seed's L-layer embedding is the full graph's:
    evaluation through the loader must reproduce the full-batch predictions (eval mode: running statistics, no
    dropout) -- row for row, in node order."""
    from stem_gnn_amd.task.node import ft_node, eval_node, _accumulate_minibatch_predictions, _run_full_batch
    from stem_gnn_amd.utils.loader import get_loader
    D, L, H, K, C = 128, 2, 4, 64, 7
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 0.0, "use_vq": 1, "setting": "standard",
              "task": "node", "num_layers": L, "batch_size": 64}
    data, labels, split = cora_like(seed=2, d=D, t=3)
    _, gm = build_pair(D, L, H, K, C, dev, params, dropout=0.0, normalize="batch")
    train_loader, subgraph_loader = get_loader(data, split, labels, params, device=dev)
    assert len(train_loader) == 3 and len(subgraph_loader) == (2708 + 511) // 512
    b = next(iter(train_loader))
    assert b.batch_size == 64 and torch.equal(b.y, labels.to(dev)[b.n_id]) and bool(split["train"].to(dev)[b.n_id[:64]].all())
    opt = torch.optim.AdamW(gm.parameters(), lr=5e-3)
    losses = [ft_node(gm, data, train_loader, opt, split, labels, params)["loss"] for _ in range(6)]
    assert losses[-1] < losses[0]
    gm.eval()
    with torch.no_grad():
        pred_l, y_l = _accumulate_minibatch_predictions(gm, subgraph_loader, dev)
        z, y_f = _run_full_batch(gm, data, labels, split, params)
        pred_f = gm.get_lin_logits(z).mean(1).softmax(dim=-1)
    assert torch.equal(y_l, y_f)
    torch.testing.assert_close(pred_l, pred_f, rtol=1e-4, atol=1e-5)
    res_l = eval_node(gm, data, subgraph_loader, split, labels, params)
    res_f = eval_node(gm, data, None, split, labels, params)
    for k in ("train", "val", "test"):
        assert abs(res_l[k] - res_f[k]) < 1e-6, (k, res_l, res_f)


def test_get_loader_link_task_full_neighbourhood_eval_equals_full_batch(dev):
    """The link loaders (reference utils/loader.py:27-46): LinkNeighborLoader [30] * L over the training edges,
    [-1] * L over every edge in batches of 4 096; seeds = the distinct endpoints of a batch's edges,
    ``edge_label_index`` in local ids.  Full-neighbourhood evaluation through the loader = full-batch evaluation."""
    from stem_gnn_amd.task.link import ft_link, eval_link
    from stem_gnn_amd.utils.loader import get_loader
    D, L, H, K, C = 128, 2, 4, 64, 5
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 0.0, "use_vq": 1, "setting": "standard",
              "task": "link", "lamda_env": 0.0, "num_layers": L, "batch_size": 1024}
    data, _, _ = cora_like(seed=4, n=2000, e=9000, d=D, c=C, t=C)
    data.x = torch.arange(2000)
    E = data.edge_index.size(1)
    labels = data.xe.clone()
    perm = torch.randperm(E, generator=torch.Generator().manual_seed(1))
    split = {}
    for name, lo, hi in (("train", 0, 4000), ("valid", 4000, 5000), ("test", 5000, 8000)):
        m = torch.zeros(E, dtype=torch.bool)
        m[perm[lo:hi]] = True
        split[name] = m
    _, gm = build_pair(D, L, H, K, C, dev, params, dropout=0.0, normalize="batch")
    train_loader, subgraph_loader = get_loader(data, split, labels, params, device=dev)
    assert len(train_loader) == 4 and len(subgraph_loader) == (E + 4095) // 4096
    b = next(iter(train_loader))
    mask = split["train"].to(dev)
    ei_tr, y_tr = data.edge_index.to(dev)[:, mask], labels.to(dev)[mask]   # what the training loader was given
    assert torch.equal(b.n_id[b.edge_label_index], ei_tr[:, b.input_id])   # local ids address the batch's nodes
    assert torch.equal(b.edge_label, y_tr[b.input_id]) and b.input_id.numel() == 1024
    assert b.batch_size == b.edge_label_index.unique().numel()             # seeds = the distinct endpoints
    opt = torch.optim.AdamW(gm.parameters(), lr=5e-3)
    losses = [ft_link(gm, data, train_loader, opt, split, labels, params)["loss"] for _ in range(4)]
    assert losses[-1] < losses[0]
    res_l = eval_link(gm, data, subgraph_loader, split, labels, params)
    res_f = eval_link(gm, data, None, split, labels, params)
    for k in ("train", "val", "test"):
        assert abs(res_l[k] - res_f[k]) <= 2 * 100.0 / int(split["valid" if k == "val" else k].sum()) + 1e-6, (k, res_l, res_f)


def test_get_loader_graph_task_batches_are_disjoint_unions(dev):
    """The graph-task loaders (reference utils/loader.py:48-72): one DataLoader per split; a batch is the disjoint
    union of its graphs (features and edge rows concatenated, endpoints shifted, ``batch`` = owner graph, ``y`` per
    graph) and runs through ft_graph / eval_graph."""
    from stem_gnn_amd.task.graph import ft_graph, eval_graph
    from stem_gnn_amd.utils.loader import get_loader
    D, L, H, K, T = 64, 2, 4, 32, 3
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 0.0, "use_vq": 1, "setting": "standard",
              "task": "graph", "lamda_env": 0.0, "num_layers": L, "batch_size": 16}
    g = torch.Generator().manual_seed(0)
    ntab = torch.nn.functional.normalize(torch.randn(40, D, generator=g), dim=-1)   # shared text rows (atom / bond kinds)
    etab = torch.nn.functional.normalize(torch.randn(5, D, generator=g), dim=-1)
    graphs = []
    for _ in range(60):
        n = int(torch.randint(6, 20, (1,), generator=g))
        e = int(torch.randint(n, 3 * n, (1,), generator=g))
        d = Data()
        d.x = torch.randint(0, 40, (n,), generator=g)
        d.xe = torch.randint(0, 5, (e,), generator=g)
        d.edge_index = torch.randint(0, n, (2, e), generator=g)
        d.node_text_feat, d.edge_text_feat = ntab, etab
        d.y = (torch.rand(T, generator=g) < 0.5).float()
        graphs.append(d)
    split = {"train": torch.arange(0, 40), "valid": torch.arange(40, 50), "test": torch.arange(50, 60)}
    tr, va, te = get_loader(graphs, split, None, params, device=dev)
    assert (len(tr), len(va), len(te)) == (3, 1, 1)
    b = next(iter(va))
    part = graphs[40:50]
    sizes = [p.x.numel() for p in part]
    assert b.node_text_feat.size(0) == sum(sizes) and tuple(b.y.shape) == (10, T) and b.batch.max().item() == 9
    off = 0
    for k, p in enumerate(part):   # graph k's rows, edges and label sit where the union puts them
        assert torch.equal(b.node_text_feat[off:off + sizes[k]].cpu(), ntab[p.x]) and torch.equal(b.y[k].cpu(), p.y)
        sel = (b.batch[b.edge_index[1]] == k).cpu()
        assert torch.equal(b.edge_index[:, sel.to(dev)].cpu() - off, p.edge_index)
        assert torch.equal(b.edge_text_feat[sel.to(dev)].cpu(), etab[p.xe])
        off += sizes[k]
    _, gm = build_pair(D, L, H, K, T, dev, params, normalize="batch", dropout=0.0)
    opt = torch.optim.AdamW(gm.parameters(), lr=2e-3)
    losses = [ft_graph(gm, None, tr, opt, None, None, params)["loss"] for _ in range(5)]
    assert losses[-1] < losses[0]
    res = eval_graph(gm, None, [tr, va, te], None, None, params)
    assert res["metric"] == "auc" and all(0.0 <= res[k] <= 100.0 for k in ("train", "val", "test"))


def test_eval_metrics_match_their_definitions(dev):
    from stem_gnn_amd.utils.eval import eval_acc, eval_auc
    torch.manual_seed(0)
    pred = torch.rand(1000, 5, device=dev)
    y = torch.randint(0, 5, (1000,), device=dev)
    mask = torch.rand(1000, device=dev) < 0.3
    assert abs(eval_acc(pred, y, mask) - (pred[mask].argmax(1) == y[mask]).float().mean().item()) < 1e-7
    # AUC against the O(n^2) definition, ties included
    s = torch.randint(0, 50, (400, 2), device=dev).float()
    t = (torch.rand(400, 2, device=dev) < 0.4).float()
    t[::7, 1] = float("nan")
    exp = []
    for i in range(2):
        v = t[:, i] == t[:, i]
        p, q = s[v, i][t[v, i] == 1], s[v, i][t[v, i] == 0]
        exp.append(((p[:, None] > q[None, :]).double().mean() + 0.5 * (p[:, None] == q[None, :]).double().mean()).item())
    assert abs(eval_auc(s, t) - sum(exp) / 2) < 1e-9


def test_link_classification_finetune_step_and_eval(dev):
    """task/link.py, full batch, on a small knowledge-graph-shaped stand-in (typed edges, relation class per edge):
    edge embedding = mean of the endpoint embeddings -> TaskModel; loss terms, parameters and predictions against
    the oracle with the dropout draws replayed."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.task.link import ft_link, eval_link
    from stem_gnn_amd.utils.others import freeze_params
    D, L, H, K, C = 128, 2, 4, 64, 11
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 0.0, "use_vq": 1, "setting": "standard",
              "task": "link", "lamda_env": 0.0}
    data, _, _ = cora_like(seed=3, n=3000, e=16000, d=D, c=C, t=C)
    data.x = torch.arange(3000)                       # span_node_and_edge_idx: one text row per node
    E = data.edge_index.size(1)
    g = torch.Generator().manual_seed(5)
    labels = data.xe.clone()                           # the relation type is the class (WN18RR / FB15K237 style)
    perm = torch.randperm(E, generator=g)
    split = {}
    for name, lo, hi in (("train", 0, 8000), ("valid", 8000, 10000), ("test", 10000, 14000)):
        m = torch.zeros(E, dtype=torch.bool)
        m[perm[lo:hi]] = True
        split[name] = m
    om, gm = build_pair(D, L, H, K, C, dev, params, normalize="batch")   # config/finetune.yaml link.*: normalize batch
    freeze_params(om.vq); freeze_params(gm.vq)
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-3)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-3)
    ops.manual_seed(4)
    N = 3000
    ea_cpu = data.edge_text_feat[data.xe]
    ids = torch.arange(32)
    gm.vq._rand_code_ids = lambda n, k, device: ids.to(device)
    for step in range(3):
        out_g = ft_link(gm, data, None, opt_g, split, labels, params)
        masks = [ops.dropout_keep_mask(N * D, 0.15, s, o, dev).view(N, D).cpu() for (s, o) in gm.encoder.last_dropout_keys]
        out_o = O.ft_link_full_batch_step(om, opt_o, data.node_text_feat, data.edge_index, ea_cpu, labels,
                                          split["train"], params, dropout_masks=masks, ortho_ids=ids)
        for k in ("act_loss", "jac_loss", "env_loss", "loss"):
            assert abs(out_g[k] - float(out_o[k])) <= 1e-4 * max(1.0, abs(float(out_o[k]))), (step, k, out_g, out_o)
    res_g = eval_link(gm, data, None, split, labels, params)
    res_o, pred_o = O.eval_link_full_batch(om, data.node_text_feat, data.edge_index, ea_cpu, labels, split)
    assert res_g["metric"] == "acc"
    for k in ("train", "val", "test"):
        assert abs(res_g[k] - res_o[k]) <= 2 * 100.0 / int(split["valid" if k == "val" else k].sum()) + 1e-6


def test_graph_level_multitask_finetune_and_auc(dev):
    """task/graph.py on molecule-shaped batches (disjoint unions of small graphs, dense per-edge text rows, a label
    matrix with missing entries): the step's losses against the oracle, and ROC-AUC evaluation runs end to end."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.task.graph import ft_graph, eval_graph
    D, L, H, K, T = 64, 2, 4, 32, 3
    params = {"separate_decoder_for_each_head": True, "decoder_jac_coeff": 1e-4, "use_vq": 1, "setting": "standard",
              "task": "graph", "lamda_env": 0.0}
    g = torch.Generator().manual_seed(0)

    def make_batch(num_graphs):
        sizes = torch.randint(8, 30, (num_graphs,), generator=g)
        n = int(sizes.sum())
        batch = torch.repeat_interleave(torch.arange(num_graphs), sizes)
        start = torch.cumsum(sizes, 0) - sizes
        src, dst = [], []
        for i in range(num_graphs):
            e = int(sizes[i]) * 2
            u = torch.randint(0, int(sizes[i]), (e,), generator=g) + int(start[i])
            v = torch.randint(0, int(sizes[i]), (e,), generator=g) + int(start[i])
            src.append(torch.cat([u, v])); dst.append(torch.cat([v, u]))
        b = Data()
        b.edge_index = torch.stack([torch.cat(src), torch.cat(dst)])
        b.node_text_feat = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=-1)
        b.edge_text_feat = torch.nn.functional.normalize(torch.randn(b.edge_index.size(1), D, generator=g), dim=-1)
        b.batch = batch
        y = torch.randint(0, 2, (num_graphs, T), generator=g).float()
        y[torch.rand(num_graphs, T, generator=g) < 0.2] = float("nan")      # missing labels
        b.y = y
        return b

    loader = [make_batch(48) for _ in range(3)]
    om, gm = build_pair(D, L, H, K, T, dev, params, normalize="batch")
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-3)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-3)
    ops.manual_seed(2)
    ids = torch.arange(32)
    gm.vq._rand_code_ids = lambda n, k, device: ids.to(device)
    # one batch at a time so the oracle can replay each batch's dropout draws
    for b in loader:
        out_g = ft_graph(gm, None, [b], opt_g, None, None, params)
        n = b.node_text_feat.size(0)
        masks = [ops.dropout_keep_mask(n * D, 0.15, s, o, dev).view(n, D).cpu() for (s, o) in gm.encoder.last_dropout_keys]
        om.train()
        z = om.encode_graph(b.node_text_feat, b.edge_index, b.edge_text_feat, b.batch, "mean", dropout_masks=masks)
        act = om.compute_activation_loss(z, b.y.clone().double(), task="multi", ortho_ids=ids)
        loss = act + om.decoder_jacobian_penalty()
        opt_o.zero_grad(); loss.backward(); opt_o.step()
        act, loss = act.detach(), loss.detach()
        assert abs(out_g["act_loss"] - float(act)) <= 1e-4 * max(1.0, abs(float(act))), (out_g, float(act))
        assert abs(out_g["loss"] - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    res = eval_graph(gm, None, [loader, loader[:1], None], None, None, params)
    assert res["metric"] == "auc" and 0.0 <= res["train"] <= 100.0 and 0.0 <= res["val"] <= 100.0
    assert res["test"] != res["test"]                                     # no loader -> NaN (task/graph.py:74-75)
