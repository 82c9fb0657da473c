"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, called through the
C ABI (stem_gnn_amd.ops -> ctypes -> libstemgnn_hip.so), against the CPU oracle on the same
seeded inputs.  Index work is compared bit-exactly; fp32 values within the tolerance written
next to each assertion (north_star: 1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import stem_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def rand_graph(n, e, seed, self_loops=True, dups=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if dups and e >= 4:
        ei[:, 1] = ei[:, 0]  # duplicate edge
    if self_loops and e >= 4:
        ei[1, 2] = ei[0, 2]  # self loop
    return ei


# ---------------------------------------------------------------------------- graph build
@pytest.mark.parametrize("n,e", [(1, 0), (7, 0), (5, 9), (1000, 5000), (4097, 100003)])
@pytest.mark.parametrize("key_row", [0, 1])
def test_csr_build_bit_exact(dev, n, e, key_row):
    from stem_gnn_amd import ops
    ei = rand_graph(n, e, seed=n * 31 + e)
    rowptr, other, eid, bad = ops.csr_build(ei.to(dev), n, key_row)
    assert int(bad.item()) == 0
    keys = ei[key_row].numpy()
    order = np.argsort(keys, kind="stable")
    exp_rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(keys, minlength=n), out=exp_rowptr[1:])
    assert rowptr.dtype == torch.int32 and other.dtype == torch.int32 and eid.dtype == torch.int32
    assert np.array_equal(rowptr.cpu().numpy(), exp_rowptr)
    assert np.array_equal(eid.cpu().numpy(), order)
    assert np.array_equal(other.cpu().numpy(), ei[1 - key_row].numpy()[order])


def test_csr_build_drops_out_of_range_edges(dev):
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    n = 10
    ei = torch.tensor([[0, 1, 12, 3, -1, 4], [1, 2, 3, 10, 2, 5]])
    rowptr, other, eid, bad = ops.csr_build(ei.to(dev), n, 1)
    assert int(bad.item()) == 3
    assert int(rowptr[-1].item()) == 3  # only the valid edges are reachable
    assert sorted(eid[:3].cpu().tolist()) == [0, 1, 5]
    with pytest.raises(IndexError):
        GraphStructure(ei.to(dev), n, validate=True)


# ---------------------------------------------------------------------------- K1 / K2
@pytest.mark.parametrize("n,e,d", [(1, 0, 32), (6, 0, 128), (13, 40, 32), (257, 3000, 48), (1000, 20000, 128),
                                   (333, 4000, 768), (50, 5000, 128), (2000, 3000, 1024)])
@pytest.mark.parametrize("mode", ["none", "dense", "table"])
def test_sage_agg_fwd_bwd(dev, n, e, d, mode):
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.model.encoder import aggregate
    torch.manual_seed(n + e + d)
    ei = rand_graph(n, e, seed=e + 7)
    x = torch.randn(n, d)
    T = 5
    table = torch.randn(T, d)
    et = torch.randint(0, T, (e,))
    ea_cpu = {"none": None, "dense": table[et], "table": table[et]}[mode]
    x_ref = x.clone().requires_grad_(True)
    ref = O.sage_mean_aggregate(x_ref, ei, ea_cpu)
    w = torch.randn(n, d)
    (ref * w).sum().backward()

    xg = x.to(dev).requires_grad_(True)
    if mode == "none":
        ea = None
    elif mode == "dense":
        ea = ea_cpu.to(dev)
    else:
        ea = EdgeTypeAttr(table.to(dev), et.to(dev))
    out = aggregate(xg, ei.to(dev), ea)
    (out * w.to(dev)).sum().backward()
    # deterministic sequential sums in edge order: agreement is ~1e-6; bar is 1e-4 (north_star)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xg.grad.cpu(), x_ref.grad, rtol=1e-5, atol=1e-5)


def test_sage_agg_known_answers(dev):
    """Hand-derived KATs: path graph means, isolated node -> 0 row, duplicate edge
    counted twice in sum and degree, self loop, relu of negative messages."""
    from stem_gnn_amd.model.encoder import aggregate
    x = torch.tensor([[1.0, -2.0, 3.0, 0.5], [2.0, 4.0, -6.0, 1.0], [-1.0, -1.0, 2.0, 2.0], [9.0, 9.0, 9.0, 9.0]])
    # edges (src->dst): 0->1, 2->1, 0->1 (dup), 2->2 (self loop); node 0 and 3 have no in-edges
    ei = torch.tensor([[0, 2, 0, 2], [1, 1, 1, 2]])
    out = aggregate(x.to(dev), ei.to(dev), None).cpu()
    r = torch.relu(x)
    exp = torch.zeros(4, 4)
    exp[1] = (r[0] + r[2] + r[0]) / 3.0
    exp[2] = r[2]
    torch.testing.assert_close(out, exp, rtol=0, atol=1e-6)
    # with an edge term that pushes some messages negative
    ea = torch.tensor([[-5.0, 0, 0, 0], [0, 0, -10.0, 0], [0, 5.0, 0, 0], [1.0, 1.0, 1.0, 1.0]])
    out = aggregate(x.to(dev), ei.to(dev), ea.to(dev)).cpu()
    exp = torch.zeros(4, 4)
    exp[1] = (torch.relu(x[0] + ea[0]) + torch.relu(x[2] + ea[1]) + torch.relu(x[0] + ea[2])) / 3.0
    exp[2] = torch.relu(x[2] + ea[3])
    torch.testing.assert_close(out, exp, rtol=0, atol=1e-6)


def test_sage_agg_hub_node(dev):
    """Skewed degrees: one hub receives every edge (G-lane chunking + tail masking)."""
    from stem_gnn_amd.model.encoder import aggregate
    n, e, d = 300, 7001, 128
    torch.manual_seed(3)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.full((e,), 17)])
    x, ea = torch.randn(n, d), torch.randn(e, d)
    ref = O.sage_mean_aggregate(x, ei, ea)
    out = aggregate(x.to(dev), ei.to(dev), ea.to(dev)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)  # 7001-term fp32 sums, different association


def skewed_graph(n, e, seed):
    """Power-law targets AND sources: a handful of hubs with thousands of in-/out-edges."""
    g = torch.Generator().manual_seed(seed)
    dst = (torch.rand(e, generator=g) ** 6 * n).long().clamp(max=n - 1)
    src = (torch.rand(e, generator=g) ** 6 * n).long().clamp(max=n - 1)
    return torch.stack([src, dst])


@pytest.mark.parametrize("n,e,d", [(400, 30000, 32), (400, 30000, 48), (1500, 60000, 128), (300, 20000, 768),
                                   (200, 9000, 1024)])
@pytest.mark.parametrize("mode", ["none", "dense", "table"])
def test_sage_agg_heavy_row_split(dev, n, e, d, mode):
    """Hubs are cut into chunk items + a combine pass (stemgnn_sage_agg_fwd/bwd_split): against the
    oracle, against the unsplit kernels (light rows: same bits), and plan reuse gives the same bits."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure
    from stem_gnn_amd.model.encoder import aggregate
    torch.manual_seed(e + d)
    ei = skewed_graph(n, e, seed=d)
    indeg, outdeg = torch.bincount(ei[1], minlength=n), torch.bincount(ei[0], minlength=n)
    assert int(indeg.max()) > 8 * ops.SPLIT_HEAVY and int(outdeg.max()) > 8 * ops.SPLIT_HEAVY
    x, w = torch.randn(n, d), torch.randn(n, d)
    table, et = torch.randn(5, d), torch.randint(0, 5, (e,))
    ea_cpu = None if mode == "none" else table[et]
    x_ref = x.clone().requires_grad_(True)
    ref = O.sage_mean_aggregate(x_ref, ei, ea_cpu)
    (ref * w).sum().backward()
    # thousands of signed fp32 terms per hub row cancel: the yardstick is the same oracle in fp64, and the HIP
    # result must be as close to it as the fp32 oracle is (chunked sums are in fact closer than sequential ones)
    x64 = x.double().requires_grad_(True)
    ref64 = O.sage_mean_aggregate(x64, ei, None if ea_cpu is None else ea_cpu.double())
    (ref64 * w.double()).sum().backward()

    def check(got, want32, want64):
        err = (got.double().cpu() - want64).abs().max().item()
        err32 = (want32.double() - want64).abs().max().item()
        assert err <= max(1.5 * err32, 1e-6 * want64.abs().max().item()), (err, err32)
        torch.testing.assert_close(got.cpu(), want32, rtol=1e-4, atol=1e-4 * want64.abs().max().item())

    ea = {"none": None, "dense": None if ea_cpu is None else ea_cpu.to(dev),
          "table": EdgeTypeAttr(table.to(dev), et.to(dev))}[mode]

    def run(gs):
        xg = x.to(dev).requires_grad_(True)
        out = aggregate(xg, gs, ea)
        (out * w.to(dev)).sum().backward()
        return out.detach(), xg.grad

    et_dev = et.to(dev) if mode == "table" else None
    gs = GraphStructure(ei.to(dev), n, et_dev)            # validation on -> degree bounds known -> split
    assert gs.max_in_degree == int(indeg.max()) and gs.max_out_degree == int(outdeg.max())
    out, gx = run(gs)
    assert gs._plan_in is not None and gs._plan_out is not None
    items, heavy = gs._plan_in.counts.tolist()
    assert heavy == int((indeg > ops.SPLIT_HEAVY).sum())
    assert items == int(((indeg[indeg > ops.SPLIT_HEAVY] + ops.SPLIT_CHUNK - 1) // ops.SPLIT_CHUNK).sum())
    check(out, ref.detach(), ref64.detach())
    check(gx, x_ref.grad, x64.grad)
    out2, gx2 = run(gs)                                                         # plan reused (build_plan = 0)
    assert torch.equal(out2, out) and torch.equal(gx2, gx)
    plain = GraphStructure(ei.to(dev), n, et_dev, validate=False)
    plain.max_in_degree = plain.max_out_degree = 0                              # bound says "no hubs": one pass
    out3, gx3 = run(plain)
    assert plain._plan_in is None and plain._plan_out is None
    light_in, light_out = (indeg <= ops.SPLIT_HEAVY).to(dev), (outdeg <= ops.SPLIT_HEAVY).to(dev)
    assert torch.equal(out3[light_in], out[light_in]) and torch.equal(gx3[light_out], gx[light_out])
    check(out3, ref.detach(), ref64.detach())
    check(gx3, x_ref.grad, x64.grad)
    unknown = GraphStructure(ei.to(dev), n, et_dev, validate=False)             # no bound -> split, same bits
    out4, gx4 = run(unknown)
    assert unknown.max_in_degree is None and torch.equal(out4, out) and torch.equal(gx4, gx)


def test_heavy_row_split_on_augmented_graph_and_plain_mean(dev):
    """dropout_undirected output (symmetric: one plan for both directions) and MixtureSageLayer's
    plain mean go through the split entry points too."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    from stem_gnn_amd.model.encoder import aggregate
    n, e, d = 600, 40000, 128
    torch.manual_seed(5)
    ei = skewed_graph(n, e, seed=1)
    x, w = torch.randn(n, d), torch.randn(n, d)
    ea = torch.randn(e, d)
    gs = GraphStructure(ei.to(dev), n)
    keep = torch.rand(e) < 0.7
    aug = gs.dropout_undirected(0.3, keep=keep.to(dev))
    ei_o, ea_o, _ = O.dropout_adj_undirected(ei, ea, keep)
    x_ref = x.clone().requires_grad_(True)
    ref = O.sage_mean_aggregate(x_ref, ei_o, ea_o)
    (ref * w).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    out = aggregate(xg, aug, ea.to(dev))
    (out * w.to(dev)).sum().backward()
    assert aug._plan_in is not None and aug._plan_out is None and int(aug._plan_in.counts[1]) > 0
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xg.grad.cpu(), x_ref.grad, rtol=1e-4, atol=1e-4 * x_ref.grad.abs().max().item())
    # plain mean (no edge term, no relu)
    xm = x.to(dev).requires_grad_(True)
    m = ops.MeanAggFn.apply(xm, gs)
    (m * w.to(dev)).sum().backward()
    xr = x.clone().requires_grad_(True)
    deg = torch.bincount(ei[1], minlength=n).clamp(min=1).unsqueeze(1)
    mr = torch.zeros(n, d).index_add_(0, ei[1], xr[ei[0]]) / deg
    (mr * w).sum().backward()
    assert gs._plan_in is not None and int(gs._plan_in.counts[1]) > 0
    torch.testing.assert_close(m.detach().cpu(), mr.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xm.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4 * xr.grad.abs().max().item())


def test_split_plan_capacity_on_one_directional_graph(dev):
    """Every edge points to a higher id, 65 out + 65 in per interior node: dropout_undirected keeps every src <= dst
    edge that survives the draw and mirrors it, so the augmented CSR holds up to 2E live slots with degrees of ~130 --
    just above SPLIT_HEAVY, where the number of chunk items per edge is largest (3 items per 130 edges).  The plan
    must be sized from the 2E slot capacity (ops.SplitPlan over GraphStructure.slot_capacity), not from the
    attribute-row count E."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    from stem_gnn_amd.model.encoder import aggregate
    n, per, d = 1200, 65, 32
    g = torch.Generator().manual_seed(11)
    src = torch.arange(n).repeat_interleave(per)
    dst = src + torch.arange(1, per + 1).repeat(n)   # node i -> i + 1 .. i + 65, where the target exists
    ok = dst < n
    ei = torch.stack([src[ok], dst[ok]])
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    e = ei.size(1)
    keep = torch.ones(e, dtype=torch.bool)
    keep[torch.randperm(e, generator=g)[:e // 200]] = False  # a few drops; the bulk stays at degree 130
    gs = GraphStructure(ei.to(dev), n)
    aug = gs.dropout_undirected(0.2, keep=keep.to(dev))
    assert aug.slot_capacity == 2 * e
    ea = torch.randn(e, d, generator=g)
    x, w = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    ei_o, ea_o, _ = O.dropout_adj_undirected(ei, ea, keep)
    deg = torch.bincount(ei_o[1], minlength=n)
    heavy = deg > ops.SPLIT_HEAVY
    assert int(heavy.sum()) > n // 2 and int(deg.max()) < 3 * ops.SPLIT_HEAVY
    x_ref = x.clone().requires_grad_(True)
    ref = O.sage_mean_aggregate(x_ref, ei_o, ea_o)
    (ref * w).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    out = aggregate(xg, aug, ea.to(dev))
    (out * w.to(dev)).sum().backward()
    plan = aug._plan_in
    items, rows = plan.counts.tolist()
    want_items = int(((deg[heavy] + ops.SPLIT_CHUNK - 1) // ops.SPLIT_CHUNK).sum())
    assert rows == int(heavy.sum()) and items == want_items
    assert items > 2 * (e // ops.SPLIT_CHUNK) + 2, "the case must exceed what a plan sized from E would hold"
    assert items <= plan.cap_items and rows <= plan.cap_heavy
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xg.grad.cpu(), x_ref.grad, rtol=1e-4, atol=1e-4 * x_ref.grad.abs().max().item())


# ---------------------------------------------------------------------------- K4
@pytest.mark.parametrize("n,d", [(2, 4), (37, 32), (1000, 128), (513, 768), (4096, 48)])
@pytest.mark.parametrize("use_bn,act,p,slope", [(True, 1, 0.15, 0.0), (True, 0, 0.0, 0.0), (False, 1, 0.3, 0.01),
                                                (True, 1, 0.0, 0.01), (False, 1, 0.0, 0.0)])
def test_bn_act_dropout_fwd_bwd(dev, n, d, use_bn, act, p, slope):
    from stem_gnn_amd import ops
    torch.manual_seed(n * 7 + d)
    y = torch.randn(n, d) * 2 + 0.5
    gamma, beta = torch.rand(d) + 0.5, torch.randn(d)
    rm, rv = torch.zeros(d), torch.ones(d)
    w = torch.randn(n, d)
    seed, offset = 1234567, 3
    keep = ops.dropout_keep_mask(n * d, p, seed, offset, dev).view(n, d).cpu()
    if p > 0:
        frac = keep.float().mean().item()
        assert abs(frac - (1 - p)) < 0.05 + 2.0 / (n * d) ** 0.5
    # reference
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    t = yr
    if use_bn:
        t = torch.nn.functional.batch_norm(t, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if act:
        t = torch.nn.functional.leaky_relu(t, slope) if slope > 0 else torch.relu(t)
    if p > 0:
        t = t * keep.float() / (1 - p)
    (t * w).sum().backward()
    # HIP
    yg = y.to(dev).requires_grad_(True)
    gg, bg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmg, rvg = rm.to(dev), rv.to(dev)
    out = ops.BnActDropFn.apply(yg, gg if use_bn else None, bg if use_bn else None, rmg if use_bn else None,
                                rvg if use_bn else None, use_bn, 0.1, 1e-5, act, slope, p, seed, offset)
    (out * w.to(dev)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), t.detach(), rtol=1e-4, atol=1e-5)
    # N == 2 is degenerate for batch norm (x_hat = +-1, rstd ~ 1/|dy|): the backward is a
    # cancellation of O(rstd) terms, so the absolute error scales with rstd there
    torch.testing.assert_close(yg.grad.cpu(), yr.grad, rtol=1e-4, atol=2e-5 if n > 2 else 2e-4)
    if use_bn:
        torch.testing.assert_close(gg.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bg.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rmg.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rvg.cpu(), rv_ref, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------- K11 / K12 / lookups / K14
@pytest.mark.parametrize("n,e,d", [(5, 0, 32), (10, 33, 32), (500, 4001, 128), (300, 1000, 768)])
def test_edge_ops(dev, n, e, d):
    from stem_gnn_amd import ops
    torch.manual_seed(e + d)
    z = torch.randn(n, d)
    ei = rand_graph(n, e, seed=5)
    zr = z.clone().requires_grad_(True)
    dot = (zr[ei[0]] * zr[ei[1]]).sum(dim=1)
    cat = torch.cat([zr[ei[0]], zr[ei[1]]], dim=-1)
    w1, w2 = torch.randn(e), torch.randn(e, 2 * d)
    ((dot * w1).sum() + (cat * w2).sum()).backward()
    zg = z.to(dev).requires_grad_(True)
    dot_g = ops.EdgeDotFn.apply(zg, ei.to(dev))
    cat_g = ops.EdgeConcatFn.apply(zg, ei.to(dev))
    ((dot_g * w1.to(dev)).sum() + (cat_g * w2.to(dev)).sum()).backward()
    torch.testing.assert_close(dot_g.detach().cpu(), dot.detach(), rtol=1e-5, atol=1e-5)
    assert torch.equal(cat_g.detach().cpu(), cat.detach())  # pure data movement: bit-exact
    torch.testing.assert_close(zg.grad.cpu(), zr.grad, rtol=1e-4, atol=1e-4)  # fp32 atomics: order-dependent


def test_gather_rows_and_ema_lerp(dev):
    from stem_gnn_amd import ops
    torch.manual_seed(0)
    table = torch.randn(1000, 128)
    idx = torch.randint(0, 1000, (4097,))
    out = ops.gather_rows(table.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), table[idx])
    t, s = torch.randn(100003), torch.randn(100003)
    tg = t.to(dev)
    ops.ema_lerp_(tg, s.to(dev), 0.99)
    torch.testing.assert_close(tg.cpu(), t * 0.99 + s * (1 - 0.99), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------- K3 / K5
@pytest.mark.parametrize("m,k,n", [(4096, 128, 128), (3000, 512, 128), (2500, 128, 512), (1111, 768, 96), (257, 36, 40)])
def test_linear_split_bf16_product_is_as_exact_as_fp32_mfma(dev, m, k, n):
    """The default dense products cut every fp32 operand exactly into three bf16 pieces (csrc/linear.hip).  Against
    an fp64 product of the same fp32 inputs -- mixed magnitudes and signs, so cancellation is real -- they must be no
    further from the truth than the fp32-MFMA kernels (mode 0), for Y = X W^T + b, dW = dY^T X and db."""
    from stem_gnn_amd import ops
    torch.manual_seed(m + k)
    x = torch.randn(m, k) * torch.exp(2.0 * torch.randn(m, k))
    w = torch.randn(n, k) * torch.exp(2.0 * torch.randn(n, k)) / k ** 0.5
    b = torch.randn(n)
    g = torch.randn(m, n) * torch.exp(2.0 * torch.randn(m, n))
    y64 = x.double() @ w.double().t() + b.double()
    dw64, db64 = g.double().t() @ x.double(), g.double().sum(0)
    xg, wg, bg, gg = x.to(dev), w.to(dev), b.to(dev), g.to(dev)
    prev = ops.linear_set_mode(-1)
    errs = {}
    try:
        for mode in (0, 1):
            ops.linear_set_mode(mode)
            y = ops.linear_fwd(xg, wg, None, None, bg, False)[0]
            dw, db = ops.linear_bwd_weight(gg, xg, True)
            errs[mode] = [((a.double().cpu() - r).abs().max() / r.abs().max()).item()
                          for a, r in ((y, y64), (dw, dw64), (db, db64))]
    finally:
        ops.linear_set_mode(prev)
    for e0, e1 in zip(errs[0], errs[1]):
        assert e1 <= max(1.5 * e0, 2e-7), (errs)
        assert e1 < 1e-5


@pytest.mark.parametrize("m,k1,k2,n,bias", [(1, 32, 0, 32, True), (130, 32, 32, 64, True), (1000, 128, 128, 128, True),
                                            (777, 128, 0, 512, True), (513, 512, 0, 128, False),
                                            (4100, 768, 768, 768, True), (300, 48, 0, 36, True)])
def test_linear_fwd_bwd_and_stats(dev, m, k1, k2, n, bias):
    from stem_gnn_amd import ops
    torch.manual_seed(m + n)
    x1, w1 = torch.randn(m, k1), torch.randn(n, k1) / k1 ** 0.5
    x2 = torch.randn(m, k2) if k2 else None
    w2 = torch.randn(n, k2) / k2 ** 0.5 if k2 else None
    b = torch.randn(n) if bias else None
    g = torch.randn(m, n)
    refs = [t.clone().requires_grad_(True) if t is not None else None for t in (x1, w1, x2, w2, b)]
    yr = refs[0] @ refs[1].t()
    if k2:
        yr = yr + refs[2] @ refs[3].t()
    if bias:
        yr = yr + refs[4]
    (yr * g).sum().backward()
    gp = [t.to(dev).requires_grad_(True) if t is not None else None for t in (x1, w1, x2, w2, b)]
    yg, partial = ops.LinearFn.apply(gp[0], gp[1], gp[2], gp[3], gp[4], True)
    (yg * g.to(dev)).sum().backward()
    # exact-fp32 MFMA chains vs ATen's blocked GEMM: agreement ~1e-6 * sum|a*b|
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-4)
    for a, r, name in zip(gp, refs, ("x1", "w1", "x2", "w2", "bias")):
        if a is not None:
            scale = float(r.grad.abs().max()) + 1e-6
            torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=1e-4, atol=1e-5 * scale + 1e-5,
                                       msg=lambda s: f"{name}: {s}")
    # fused column statistics == statistics of the stored output
    ps = partial.sum(dim=0).cpu()
    torch.testing.assert_close(ps[0], yr.detach().sum(dim=0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(ps[1], (yr.detach() ** 2).sum(dim=0), rtol=1e-4, atol=1e-3)


# ---------------------------------------------------------------------------- augmentation / sampling
@pytest.mark.parametrize("n,e,p", [(1, 0, 0.2), (9, 30, 0.0), (50, 400, 0.2), (3000, 40000, 0.5), (100, 3000, 1.0),
                                   (256, 900, 0.1), (257, 900, 0.1), (70001, 200000, 0.3)])  # 1, 2 and 274 count blocks
@pytest.mark.parametrize("typed", [False, True])
def test_dropout_undirected_matches_pyg_semantics(dev, n, e, p, typed):
    """The fused CSR->CSR augmentation equals dropout_adj(force_undirected=True) restated in
    the oracle (same Bernoulli draw): identical multiset of (src, dst, original edge id) per
    target row IN ORDER, for both CSR views; bit-exact (index work)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    ei = rand_graph(n, e, seed=e + 3)
    et = torch.randint(0, 5, (e,))
    g = GraphStructure(ei.to(dev), n, et.to(dev) if typed else None).ensure_transpose()
    ga = g.dropout_undirected(p)
    keep = ops.dropout_keep_mask(e, p, *ga.keep_key, dev).cpu() if e else torch.zeros(0, dtype=torch.bool)
    aug_ei, _, m = O.dropout_adj_undirected(ei, None, keep)
    sel = m.nonzero().view(-1)
    orig_id = torch.cat([sel, sel])
    live = int(ga.rowptr[-1].item())
    assert live == aug_ei.size(1)
    for key_row, (rowptr, other, eid, ety) in ((1, (ga.rowptr, ga.src, ga.eid, ga.etype_slot)),
                                                (0, (ga.rowptr_t, ga.dst_t, ga.eid_t, ga.etype_slot_t))):
        order = np.argsort(aug_ei[key_row].numpy(), kind="stable")
        exp_rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(aug_ei[key_row].numpy(), minlength=n), out=exp_rowptr[1:])
        assert np.array_equal(rowptr.cpu().numpy(), exp_rowptr)
        assert np.array_equal(other[:live].cpu().numpy(), aug_ei[1 - key_row].numpy()[order])
        assert np.array_equal(eid[:live].cpu().numpy(), orig_id.numpy()[order])
        if typed:
            assert np.array_equal(ety[:live].cpu().numpy(), et[orig_id].numpy()[order])
    deg = np.diff(ga.rowptr.cpu().numpy())
    torch.testing.assert_close(ga.inv_deg.cpu(), torch.from_numpy(1.0 / np.maximum(deg, 1)).float())
    # injected keep mask gives the same graph
    gb = g.dropout_undirected(p, keep=keep.to(dev))
    assert torch.equal(gb.rowptr, ga.rowptr) and torch.equal(gb.src[:live], ga.src[:live])


def test_sampler_and_augmentation_on_graphs_without_edges(dev):
    """Seeds whose nodes have no in-edges: the batch is the seeds alone, both CSR views are empty, nothing is expanded
    (active_rows = 0 of B rows -> the augmentation's count launch is skipped), and the step-side helpers accept it."""
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    n, d = 300, 16
    ei = torch.zeros(2, 0, dtype=torch.int64, device=dev)
    xe = torch.zeros(0, dtype=torch.int64, device=dev)
    feat = torch.randn(n, d, device=dev)
    s = HipNeighborSampler(ei, xe, n, torch.arange(n, device=dev), feat, torch.randn(1, d, device=dev), [4, 3], seed=2)
    seeds = torch.tensor([5, 17, 5, 299], device=dev)
    b = s.sample(seeds)
    assert torch.equal(b.n_id, seeds) and tuple(b.edge_index.shape) == (2, 0) and b.xe.numel() == 0
    g = b.graph
    assert g.num_nodes == 4 and g.num_edges == 0 and g.active_rows in (None, 0, 4)
    assert torch.equal(g.rowptr, torch.zeros(5, dtype=torch.int32, device=dev)) and torch.equal(g.rowptr_t, g.rowptr)
    assert torch.equal(g.inv_deg, torch.ones(4, device=dev))
    a = g.dropout_undirected(0.3)
    assert int(a.rowptr[-1]) == 0 and torch.equal(a.rowptr, g.rowptr) and torch.equal(a.inv_deg, g.inv_deg)
    assert int((s.local_of != -2 ** 31).sum()) == 0
    # ... and a graph whose only edges leave the seeds (nothing points at them): same batch
    ei2 = torch.stack([torch.tensor([5, 17, 299], device=dev), torch.tensor([1, 2, 3], device=dev)])
    s2 = HipNeighborSampler(ei2, torch.zeros(3, dtype=torch.int64, device=dev), n, torch.arange(n, device=dev), feat,
                            torch.randn(1, d, device=dev), [4, 3], seed=2)
    b2 = s2.sample(seeds)
    assert torch.equal(b2.n_id, seeds) and b2.edge_index.size(1) == 0


def test_dropout_undirected_on_a_sampled_batch_examines_the_expanded_rows_only(dev):
    """A sampler batch promises that rows >= active_rows have no in-edges; the augmentation then looks at the leading
    rows only.  Same result as without the promise (every row examined), array for array."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import GraphStructure
    g = make_graph(20000, 300000, 16, 4, kind="Z", device=dev, graph_seed=2)
    s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [6, 5], seed=1)
    b = s.sample(torch.randperm(g.num_nodes, device=dev)[:300])
    ga, nb = b.graph, b.n_id.numel()
    assert ga.active_rows is not None and 300 < ga.active_rows < nb
    plain = GraphStructure(b.edge_index.clone(), nb, ga.etype_slot.clone()).ensure_transpose()
    assert plain.active_rows is None
    keep = torch.rand(ga.num_edges, device=dev) > 0.25
    a, c = ga.dropout_undirected(0.25, keep=keep), plain.dropout_undirected(0.25, keep=keep)
    live = int(c.rowptr[-1])
    assert live > 0 and torch.equal(a.rowptr, c.rowptr) and torch.equal(a.inv_deg, c.inv_deg)
    for name in ("src", "eid", "etype_slot", "dst_t", "eid_t", "etype_slot_t"):
        assert torch.equal(getattr(a, name)[:live], getattr(c, name)[:live]), name


def test_negative_sample_properties(dev):
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    n, e, k = 40, 600, 300   # dense enough that rejections actually happen
    ei = rand_graph(n, e, seed=9)
    g = GraphStructure(ei.to(dev), n)
    perm = torch.randperm(e)[:k]
    sel = torch.zeros(e, dtype=torch.uint8)
    sel[perm] = 1
    neg = ops.negative_sample(g, sel.to(dev), k, 5, 7).cpu()
    assert tuple(neg.shape) == (2, k) and neg.dtype == torch.int64
    assert O.check_negative_edges(neg, ei[:, perm], n)
    neg2 = ops.negative_sample(g, sel.to(dev), k, 5, 7).cpu()
    assert torch.equal(neg, neg2)  # pure function of (seed, offset)


def test_edge_bce_loss(dev):
    from stem_gnn_amd import ops
    n, d, kp, kn = 200, 64, 301, 257
    torch.manual_seed(4)
    z = torch.randn(n, d) * 0.3
    ei = torch.randint(0, n, (2, kp + kn))
    zr = z.clone().requires_grad_(True)
    val = (zr[ei[0]] * zr[ei[1]]).sum(1)
    ref = -torch.log(torch.sigmoid(val[:kp]) + O.EPS).mean() - torch.log(1 - torch.sigmoid(val[kp:]) + O.EPS).mean()
    (ref * 1.7).backward()
    zg = z.to(dev).requires_grad_(True)
    out = ops.EdgeBceLossFn.apply(zg, ei.to(dev), kp)
    (out * 1.7).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(zg.grad.cpu(), zr.grad, rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------- neighbour sampler
def _check_sampler_views(b, g, dev):
    """What the sampler hands over beside the by-target CSR equals what the consumers used to derive from the COO: the
    by-source CSR of a stable sort by source (rows in edge order), the types gathered through it, 1 / in-degree, and
    the int64 id / type / feature-row vectors."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    gs, nb = b.graph, b.n_id.numel()
    assert b.edge_index.is_contiguous() and b.edge_index.dtype == torch.int64
    if gs.rowptr_t is None:  # graphs with long out-rows: the sampler leaves the by-source view to a sort on demand
        gs.ensure_transpose()
    ref = GraphStructure(b.edge_index.clone(), nb, gs.etype_slot.clone()).ensure_transpose()
    for name in ("rowptr", "src", "rowptr_t", "dst_t", "eid_t", "etype_slot_t", "inv_deg"):
        assert torch.equal(getattr(gs, name), getattr(ref, name)), name
    assert torch.equal(gs.eid.long(), torch.arange(gs.num_edges, device=dev))
    assert gs.ensure_transpose().rowptr_t is gs.rowptr_t  # nothing left to build
    assert torch.equal(b.x, g.x[b.n_id]) and torch.equal(b.xe, gs.etype_slot.long())
    assert b.n_id.dtype == torch.int64 and b.xe.dtype == torch.int64


@pytest.mark.parametrize("seed", range(6))
def test_sampler_randomised_graphs_fanouts_and_seed_lists(dev, seed):
    """Random small graphs (uniform / skewed, with self-loops and repeated edges), random fan-out lists (1-3 hops, values
    up to 32 and -1) and seed lists with repeats: every batch edge is a real edge with its type, every expanded node got
    min(in-degree, fan-out) of them, nodes are numbered hop by hop, the CSR views agree with the COO, and the scratch
    map comes back clean."""
    from collections import Counter
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    rng = torch.Generator().manual_seed(100 + seed)
    n = int(torch.randint(50, 3000, (1,), generator=rng))
    e = int(torch.randint(n, 12 * n, (1,), generator=rng))
    src = torch.randint(0, n, (e,), generator=rng)
    dst = (torch.rand(e, generator=rng) ** (1 + seed % 3) * n).long().clamp(max=n - 1)  # seed % 3 > 0: skewed in-degrees
    src[:5] = dst[:5]                      # self-loops
    src[5:10], dst[5:10] = src[0], dst[1]  # one edge five times over
    ei = torch.stack([src, dst]).to(dev)
    xe = torch.randint(0, 3, (e,), generator=rng).to(dev)
    feat = torch.randn(n, 8, device=dev)
    hops = 1 + seed % 3
    fan = [int(f) for f in torch.randint(1, 33, (hops,), generator=rng)]
    if seed % 2:
        fan[int(torch.randint(0, hops, (1,), generator=rng))] = -1
    s = HipNeighborSampler(ei, xe, n, torch.arange(n, device=dev), feat, torch.randn(3, 8, device=dev), fan, seed=seed)
    indeg = torch.bincount(dst, minlength=n)
    full = Counter(zip(src.tolist(), dst.tolist(), xe.cpu().tolist()))
    for rep in range(3):
        b_sz = int(torch.randint(1, 200, (1,), generator=rng))
        seeds = torch.randint(0, n, (b_sz,), generator=rng).to(dev)  # repeats are likely
        b = s.sample(seeds)
        n_id, bei, bxe = b.n_id.cpu(), b.edge_index.cpu(), b.xe.cpu()
        nb = n_id.numel()
        assert torch.equal(n_id[:b_sz], seeds.cpu())
        assert n_id[b_sz:].unique().numel() == nb - b_sz and not bool(torch.isin(n_id[b_sz:], seeds.cpu()).any())
        got = Counter(zip(n_id[bei[0]].tolist(), n_id[bei[1]].tolist(), bxe.tolist()))
        cnt = torch.bincount(bei[1], minlength=nb)
        # a node is expanded in the hop after the one that reached it: walk the hops through the counts
        lo, hi = 0, b_sz
        for f in fan:
            lim = f if f >= 0 else 10 ** 9
            assert torch.equal(cnt[lo:hi], torch.clamp(indeg[n_id[lo:hi]], max=lim)), (fan, lo, hi)
            reached = bei[0][(bei[1] >= lo) & (bei[1] < hi)]
            lo, hi = hi, max(hi, int(reached.max()) + 1 if reached.numel() else hi)
        assert int(cnt[lo:].sum()) == 0 or lo == hi
        # repeated seeds draw separately, so an edge may appear once per copy of its target: multiplicity bound
        mult = Counter(seeds.cpu().tolist())
        assert all(c <= full[k] * max(mult.get(k[1], 1), 1) for k, c in got.items())
        gs = b.graph
        if gs.rowptr_t is None:
            gs.ensure_transpose()
        order = torch.argsort(bei[0], stable=True)
        assert torch.equal(gs.dst_t.cpu().long(), bei[1][order]) and torch.equal(gs.eid_t.cpu().long(), order)
        assert int((s.local_of != -2 ** 31).sum()) == 0


def test_sampler_plain_entry_point_gives_the_same_batch(dev):
    """stemgnn_sample_batch (by-target CSR + COO with row stride cap_edges) and stemgnn_sample_batch_views draw the same
    batch for the same (seed, offset)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    from stem_gnn_amd.data.synthetic import make_graph
    g = make_graph(3000, 40000, 16, 4, kind="U", device=dev, graph_seed=5)  # uniform: short out-rows, by-source view built
    for fan in ([4, 3], [6], [3, 2, 2]):
        s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, fan, seed=3)
        assert s.batch_max_out_degree <= 128
        seeds = torch.randperm(g.num_nodes, device=dev)[:50]
        seeds[7] = seeds[3]  # a duplicate seed keeps its first position
        n_id, rowptr, src, etype, coo, nb, eb, ab = ops.sample_batch(s.rowptr, s.src, s.etype, s.num_nodes, seeds, fan,
                                                                     s.seed, 64, s.local_of)
        b = s.sample(seeds)  # first call: offset 64
        assert b.graph.rowptr_t is not None
        assert (nb, eb, ab) == (b.n_id.numel(), b.edge_index.size(1), b.graph.active_rows if b.graph.active_rows is not None else nb)
        assert torch.equal(n_id.long(), b.n_id) and torch.equal(rowptr, b.graph.rowptr) and torch.equal(src, b.graph.src)
        assert torch.equal(etype, b.graph.etype_slot) and torch.equal(coo, b.edge_index)
        _check_sampler_views(b, g, dev)
        assert int((s.local_of != -2 ** 31).sum()) == 0
    # a caller's output arrays need not start on 16-byte boundaries (the scans take 16-byte loads where they may)
    import ctypes
    from stem_gnn_amd._lib import lib, check
    fan = [4, 3]
    s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, fan, seed=3)
    seeds = torch.randperm(g.num_nodes, device=dev)[:50]
    ref = s.sample(seeds)
    cn, ce = 50 * (1 + 4 + 12), 50 * (4 + 12)
    i32 = lambda n: torch.empty(n + 1, dtype=torch.int32, device=dev)[1:]  # 4 bytes past an aligned start
    n_id, rp, src, ty, rpt, dst_t, eid_t, ty_t = i32(cn), i32(cn + 1), i32(ce), i32(ce), i32(cn + 1), i32(ce), i32(ce), i32(ce)
    inv = torch.empty(cn + 1, dtype=torch.float32, device=dev)[1:]
    coo = torch.empty(2 * ce, dtype=torch.int64, device=dev)
    counts = torch.empty(3, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.stemgnn_sampler_workspace_bytes(50, 2, 4)), dtype=torch.uint8, device=dev)
    check(lib.stemgnn_sample_batch_views(
        s.rowptr.data_ptr(), s.src.data_ptr(), s.etype.data_ptr(), s.num_nodes, seeds.data_ptr(), 50,
        (ctypes.c_int32 * 2)(*fan), 2, s.seed, 64, s.local_of.data_ptr(), cn, ce, n_id.data_ptr(), rp.data_ptr(),
        src.data_ptr(), ty.data_ptr(), coo.data_ptr(), counts.data_ptr(), rpt.data_ptr(), dst_t.data_ptr(),
        eid_t.data_ptr(), ty_t.data_ptr(), inv.data_ptr(), None, None, None, None, ws.data_ptr(), ws.numel(),
        torch.cuda.current_stream().cuda_stream), "sample_batch_views")
    nb, eb, _ = counts.tolist()
    assert (nb, eb) == (ref.n_id.numel(), ref.edge_index.size(1))
    assert torch.equal(rp[:nb + 1], ref.graph.rowptr) and torch.equal(rpt[:nb + 1], ref.graph.rowptr_t)
    assert torch.equal(dst_t[:eb], ref.graph.dst_t) and torch.equal(eid_t[:eb], ref.graph.eid_t)
    assert torch.equal(coo[:2 * eb].view(2, eb), ref.edge_index) and torch.equal(inv[:nb], ref.graph.inv_deg)


@pytest.mark.parametrize("fan", [[5, 3], [-1, -1], [4, -1]])
@pytest.mark.parametrize("impl", ["hip", "torch"])
def test_neighbor_sampler_contract(dev, impl, fan):
    """NeighborLoader contract (reference pretrain.py:151-153): per hop each newly reached node
    draws min(deg, fanout) of its in-neighbours without replacement; every sampled edge is a real
    edge of the full graph with its edge type; seeds first; last-hop nodes have no in-edges."""
    from collections import Counter
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    from stem_gnn_amd.data.synthetic import make_graph
    from torch_sampler import NeighborSampler  # tests/torch_sampler.py: the torch-op restatement of the contract
    g = make_graph(5000, 60000, 16, 4, kind="Z", device=dev, graph_seed=3)
    cls = HipNeighborSampler if impl == "hip" else NeighborSampler
    lim = [f if f >= 0 else 10 ** 9 for f in fan]  # -1: every in-neighbour (the evaluation loaders, utils/loader.py:18-25)
    s = cls(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, fan, seed=11)
    ei, xe = g.edge_index.cpu(), g.xe.cpu()
    full = Counter(zip(ei[0].tolist(), ei[1].tolist(), xe.tolist()))
    indeg = torch.bincount(ei[1], minlength=g.num_nodes)
    seeds = torch.randperm(g.num_nodes, device=dev)[:64]
    for rep in range(3):
        b = s.sample(seeds)
        n_id, bei, bxe = b.n_id.cpu(), b.edge_index.cpu(), b.xe.cpu()
        nb, eb = n_id.numel(), bei.size(1)
        assert torch.equal(n_id[:64], seeds.cpu()) and n_id.unique().numel() == nb
        assert int(bei.min()) >= 0 and int(bei.max()) < nb
        # every sampled edge is a distinct real edge (multi-edges respected) with the right type
        got = Counter(zip(n_id[bei[0]].tolist(), n_id[bei[1]].tolist(), bxe.tolist()))
        assert all(full[k] >= c for k, c in got.items())
        # expanded nodes (dst side) received exactly min(deg, fanout) edges; hop structure
        cnt = torch.bincount(bei[1], minlength=nb)
        hop1_end = 64 + int((cnt[:64]).sum() * 0 + (bei[0][bei[1] < 64].unique() >= 64).sum())
        assert torch.equal(cnt[:64], torch.minimum(indeg[n_id[:64]], torch.tensor(lim[0])))
        assert torch.equal(cnt[64:hop1_end], torch.minimum(indeg[n_id[64:hop1_end]], torch.tensor(lim[1])))
        assert int(cnt[hop1_end:].sum()) == 0
        if impl == "hip":
            gs = b.graph
            assert gs.num_nodes == nb and gs.num_edges == eb
            exp_rowptr = torch.zeros(nb + 1, dtype=torch.int64)
            exp_rowptr[1:] = torch.cumsum(cnt, 0)
            assert torch.equal(gs.rowptr.cpu().long(), exp_rowptr)
            assert torch.equal(gs.src.cpu().long(), bei[0]) and torch.equal(gs.etype_slot.cpu().long(), bxe)
            assert bool((bei[1][1:] >= bei[1][:-1]).all())  # edge j == CSR slot j
            _check_sampler_views(b, g, dev)
            if fan[0] < 0 and fan[1] < 0 and rep == 0:
                # with every in-neighbour taken nothing is drawn: the batch is a function of the seeds alone.  The torch
                # restatement numbers a hop's new nodes by global id, this sampler by first appearance: the same nodes
                # per hop and the same edges, up to that relabelling
                t = NeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, fan, seed=1)
                bt = t.sample(seeds)
                tn, te, tx = bt.n_id.cpu(), bt.edge_index.cpu(), bt.xe.cpu()
                assert tn.numel() == nb and te.size(1) == eb
                assert torch.equal(tn[:64], n_id[:64]) and torch.equal(tn[64:hop1_end].sort().values, n_id[64:hop1_end].sort().values)
                assert torch.equal(tn[hop1_end:].sort().values, n_id[hop1_end:].sort().values)
                assert Counter(zip(tn[te[0]].tolist(), tn[te[1]].tolist(), tx.tolist())) == got
    # the scratch map is left clean and draws are uniform: over repeated draws of one high-degree
    # node every in-neighbour slot is picked with frequency ~ fanout / deg
    if impl == "hip":
        assert int((s.local_of != -2 ** 31).sum()) == 0
    if impl == "hip" and min(fan) > 0:
        v = int(indeg.argmax())
        deg = int(indeg[v])
        hits = Counter()
        reps = 400
        for _ in range(reps):
            b = s.sample(torch.tensor([v], device=dev))
            first = b.edge_index[:, b.edge_index[1] == 0]
            for u in b.n_id[first[0]].tolist():
                hits[u] += 1
        nbrs = Counter(ei[0][ei[1] == v].tolist())
        exp = {u: reps * fan[0] * m / deg for u, m in nbrs.items()}
        chi2 = sum((hits[u] - e) ** 2 / e for u, e in exp.items())
        assert chi2 < 3.0 * len(exp) + 50, (chi2, len(exp))


@pytest.mark.parametrize("n,k", [(1, 1), (5, 5), (100, 10), (111897, 11189), (128, 32), (1000003, 1000)])
def test_sample_subset_is_distinct_in_range_and_spread(dev, n, k):
    from stem_gnn_amd import ops
    a = ops.sample_subset(n, k, dev, key=(123, 1)).cpu()
    assert a.dtype == torch.int64 and a.numel() == k
    assert int(a.min()) >= 0 and int(a.max()) < n and a.unique().numel() == k  # without replacement
    assert torch.equal(a, ops.sample_subset(n, k, dev, key=(123, 1)).cpu())   # pure function of the key
    b = ops.sample_subset(n, k, dev, key=(123, 2)).cpu()
    if n >= 100 and k < n:
        assert not torch.equal(a, b)
    if n == 111897:
        # inclusion frequency over many keys ~ k/n for every element bucket (uniform subsets)
        hits = torch.zeros(n)
        reps = 200
        for r in range(reps):
            hits[ops.sample_subset(n, k, dev, key=(7, r)).cpu()] += 1
        buckets = hits.view(-1)[: (n // 100) * 100].view(100, -1).sum(1)
        exp = reps * k / n * (n // 100)
        assert float((buckets - exp).abs().max()) < 6 * exp ** 0.5


def test_mask_columns(dev):
    from stem_gnn_amd import ops
    x = torch.randn(333, 64)
    out, key = ops.mask_columns(x.to(dev), 0.3, key=(5, 9))
    keep = ops.dropout_keep_mask(64, 0.3, *key, dev).cpu()
    assert 0 < int(keep.sum()) < 64
    assert torch.equal(out.cpu(), O.mask_feature_col(x, keep))


# ---------------------------------------------------------------------------- fused scalar losses
@pytest.mark.parametrize("shape", [(1, 4), (1024, 128), (11189, 128), (37, 33)])
def test_mse_loss_fn(dev, shape):
    from stem_gnn_amd import ops
    torch.manual_seed(shape[0])
    p, t = torch.randn(*shape), torch.randn(*shape)
    pr = p.clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(pr, t)
    (ref * 3.0).backward()
    pg = p.to(dev).requires_grad_(True)
    out = ops.MseLossFn.apply(pg, t.to(dev))
    (out * 3.0).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(pg.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("rows,d", [(1, 8), (1024, 128), (200, 768), (77, 48)])
def test_cosine_loss_fn(dev, rows, d):
    from stem_gnn_amd import ops
    torch.manual_seed(rows + d)
    z, h = torch.randn(rows, d), torch.randn(rows, d)
    if rows > 2:
        h[1] = 0.0  # exercises the eps clamp of F.normalize
    hr = h.clone().requires_grad_(True)
    zn = torch.nn.functional.normalize(z, dim=-1, p=2)
    hn = torch.nn.functional.normalize(hr, dim=-1, p=2)
    ref = (1 - (zn * hn).sum(dim=-1)).mean()
    (ref * 2.0).backward()
    hg = h.to(dev).requires_grad_(True)
    out = ops.CosineLossFn.apply(z.to(dev), hg)
    (out * 2.0).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    mask = torch.ones(rows, dtype=torch.bool)
    if rows > 2:
        mask[1] = False  # d/dh at h = 0 is 1/eps * z in both; compared separately below with a relative bound
        torch.testing.assert_close(hg.grad.cpu()[1], hr.grad[1], rtol=1e-3, atol=0)
    torch.testing.assert_close(hg.grad.cpu()[mask], hr.grad[mask], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("H,K,Dc,M", [(1, 8, 16, 8), (4, 128, 128, 32), (4, 128, 768, 32), (2, 40, 48, 17)])
def test_ortho_loss_fn(dev, H, K, Dc, M):
    from stem_gnn_amd import ops
    torch.manual_seed(K + Dc)
    embed = torch.randn(H, K, Dc) * 0.7
    ids = torch.randperm(K)[:M]
    er = embed.clone().requires_grad_(True)
    ref = O.orthogonal_loss(er[:, ids]) * 1.5
    (ref * 2.0).backward()
    eg = embed.to(dev).requires_grad_(True)
    out = ops.OrthoLossFn.apply(eg, ids.to(dev), 1.5)
    (out * 2.0).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(eg.grad.cpu(), er.grad, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("scale", [3.0, 1e-3])
def test_clip_grad_norm_matches_torch(dev, scale):
    """stemgnn_clip_grad_norm against torch.nn.utils.clip_grad_norm_ (reference pretrain.py:62): clipping case and
    the no-op case (total norm below max_norm leaves every bit alone), mixed tensor sizes incl. empty and 1-element."""
    from stem_gnn_amd import ops
    torch.manual_seed(0)
    shapes = [(128, 256), (128,), (1,), (0,), (512, 128), (4, 128, 128), (100003,)]
    ps = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
    for p, q in zip(ps, qs):
        g = torch.randn(p.shape, device=dev) * scale
        p.grad, q.grad = g.clone(), g.clone()
    ps.append(torch.nn.Parameter(torch.zeros(3, device=dev)))      # no grad: skipped by both
    qs.append(torch.nn.Parameter(torch.zeros(3, device=dev)))
    t_ref = torch.nn.utils.clip_grad_norm_(qs, 1.0)
    t = ops.clip_grad_norm_(ps, 1.0)
    torch.testing.assert_close(t, t_ref, rtol=1e-5, atol=0)
    for p, q in zip(ps[:-1], qs[:-1]):
        if scale < 1:
            assert torch.equal(p.grad, q.grad)
        else:
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-9)


def test_sample_edges_matches_subset_and_indexing(dev):
    """stemgnn_sample_edges == sample_subset picks + edge_index[:, perm] + type[perm] + membership mask; the strided
    negative sampler writes the same pairs as the plain one."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    torch.manual_seed(0)
    n, E, k = 500, 7001, 700
    ei = torch.randint(0, n, (2, E), device=dev)
    et = torch.randint(0, 9, (E,), device=dev)
    key = (77, 1234)
    perm, sel, sel_type, selected = ops.sample_edges(ei, et, k, want_selected=True, pad_columns=k, key=key)
    ref = ops.sample_subset(E, k, dev, key=key)
    assert torch.equal(perm, ref) and perm.unique().numel() == k
    assert torch.equal(sel[:, :k], ei[:, perm]) and torch.equal(sel_type, et[perm])
    exp = torch.zeros(E, dtype=torch.uint8, device=dev)
    exp[perm] = 1
    assert torch.equal(selected, exp)
    g = GraphStructure(ei, n)
    neg = ops.negative_sample(g, selected, k, 5, 6)
    ops.negative_sample_into(g, selected, k, 5, 6, sel, k)
    assert torch.equal(sel[:, k:], neg) and torch.equal(sel[:, :k], ei[:, perm])
    perm2, sel2, t2, m2 = ops.sample_edges(ei, None, 1, key=key)     # k = 1, no types, no mask
    assert t2 is None and m2 is None and torch.equal(sel2[:, 0], ei[:, perm2[0]])


def test_fused_adamw_matches_torch_adamw(dev):
    """ops.FusedAdamW (+ the clipping factor folded into its read of the gradients) against
    clip_grad_norm_ + torch.optim.AdamW over several steps, lr changed between steps like the scheduler does."""
    from stem_gnn_amd import ops
    torch.manual_seed(0)
    shapes = [(128, 256), (128,), (1,), (512, 128), (4, 64, 32), (70001,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = ops.FusedAdamW(ps, lr=1e-2, weight_decay=1e-2)
    ref = torch.optim.AdamW(qs, lr=1e-2, weight_decay=1e-2)
    for step in range(6):
        scale = 3.0 if step % 2 == 0 else 1e-3          # clipping active / inactive
        for p, q in zip(ps, qs):
            g = torch.randn(p.shape, device=dev) * scale
            p.grad, q.grad = g.clone(), g.clone()
        lr = 1e-2 * (1 + step) / 4
        for o in (opt, ref):
            for grp in o.param_groups:
                grp["lr"] = lr
        out = ops.grad_norm_coef([p.grad for p in ps], 1.0)
        opt.step(grad_coef=out[1:])
        total = torch.nn.utils.clip_grad_norm_(qs, 1.0)
        ref.step()
        torch.testing.assert_close(out[0], total, rtol=1e-5, atol=0)
        for p, q in zip(ps, qs):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-5, atol=2e-6)
    for p, q in zip(ps, qs):
        torch.testing.assert_close(opt.state[p]["exp_avg"], ref.state[q]["exp_avg"], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(opt.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"], rtol=1e-4, atol=1e-9)


def test_csr_round_trip_properties_hypothesis(dev):
    """Property test (SURVEY §4 'indexing'): for arbitrary small COO lists -- duplicates, self loops, isolated nodes,
    empty -- both CSR views are stable groupings whose permutation inverts back to the input, and the aggregation over
    them equals a dense adjacency product."""
    from hypothesis import given, settings, strategies as st
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import GraphStructure
    from stem_gnn_amd.model.encoder import aggregate

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 40).flatmap(lambda n: st.tuples(st.just(n), st.lists(
        st.tuples(st.integers(0, n - 1), st.integers(0, n - 1)), min_size=0, max_size=120))))
    def check(case):
        n, edges = case
        ei = torch.tensor(edges, dtype=torch.int64).reshape(-1, 2).t().contiguous()
        e = ei.size(1)
        for key_row in (1, 0):
            rowptr, other, eid, bad = ops.csr_build(ei.to(dev), n, key_row)
            assert int(bad.item()) == 0
            rp, ot, pm = rowptr.cpu().long(), other.cpu().long(), eid.cpu().long()
            assert rp[0] == 0 and rp[-1] == e and bool((rp[1:] >= rp[:-1]).all())
            assert sorted(pm.tolist()) == list(range(e))
            keys = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
            assert torch.equal(ei[key_row][pm], keys) and torch.equal(ei[1 - key_row][pm], ot)
            same = keys[1:] == keys[:-1]
            assert bool((pm[1:][same] > pm[:-1][same]).all())
        x = torch.randn(n, 8)
        adj = torch.zeros(n, n)
        if e:
            adj.index_put_((ei[1], ei[0]), torch.ones(e), accumulate=True)
        ref = adj @ torch.relu(x) / adj.sum(1, keepdim=True).clamp(min=1)
        out = aggregate(x.to(dev), GraphStructure(ei.to(dev), n), None).cpu()
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)

    check()


@pytest.mark.parametrize("m,rows,k1,k2,n", [(1000, 130, 128, 128, 128), (1000, 0, 64, 32, 96), (777, 776, 32, 64, 64),
                                            (4100, 129, 128, 128, 512), (300, 300, 32, 32, 32)])
def test_linear_with_zero_tail_promise(dev, m, rows, k1, k2, n):
    """x1_rows: rows >= x1_rows of the first operand are zero (aggregate of a sampled batch).  Same y / statistics as
    the plain call; gradients of x1 for the promised-non-zero rows, of w1 contracted over those rows, of x2 / w2 /
    bias over all rows -- against torch on the zero-padded operand."""
    from stem_gnn_amd import ops
    torch.manual_seed(m + rows)
    x1 = torch.randn(m, k1)
    x1[rows:] = 0
    x2, w1, w2, b = torch.randn(m, k2), torch.randn(n, k1) / 8, torch.randn(n, k2) / 8, torch.randn(n)
    g = torch.randn(m, n)
    ref = [t.clone().requires_grad_(True) for t in (x1, w1, x2, w2, b)]
    yr = ref[0] @ ref[1].t() + ref[2] @ ref[3].t() + ref[4]
    (yr * g).sum().backward()
    gp = [t.to(dev).requires_grad_(True) for t in (x1, w1, x2, w2, b)]
    y, partial = ops.LinearFn.apply(gp[0], gp[1], gp[2], gp[3], gp[4], True, rows)
    y0, partial0 = ops.LinearFn.apply(gp[0].detach(), gp[1].detach(), gp[2].detach(), gp[3].detach(), gp[4].detach(), True)
    assert torch.equal(y, y0) or float((y - y0).abs().max()) < 1e-6   # skipped chunks only ever added exact zeros
    torch.testing.assert_close(partial.sum(0), partial0.sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-4)
    (y * g.to(dev)).sum().backward()
    torch.testing.assert_close(gp[0].grad[:rows].cpu(), ref[0].grad[:rows], rtol=1e-4, atol=1e-4)
    for i in (1, 2, 3, 4):
        torch.testing.assert_close(gp[i].grad.cpu(), ref[i].grad, rtol=1e-4, atol=2e-3 if i in (1, 3) else 1e-3)


# ---------------------------------------------------------------------------- round-2 kernels
@pytest.mark.parametrize("pair", [0, 1], ids=["bf16-pieces", "pair"])
@pytest.mark.parametrize("m,k2,n,rows,store", [(1000, 0, 128, -1, -1), (777, 0, 512, -1, 300), (2000, 128, 128, 300, -1),
                                               (1300, 128, 256, 0, -1), (4097, 0, 128, -1, 1000), (129, 0, 96, -1, -1),
                                               (40000, 128, 128, 4100, -1), (40000, 128, 256, 33000, 2000),
                                               (70001, 128, 128, 1, 64)])
def test_weight_stationary_product_returns_the_tile_kernels_bits(dev, m, k2, n, rows, store, pair):
    """The weight-stationary kernels (weights as register-resident matrix-core fragments, persistent blocks) against the
    tile kernel of csrc/linear.hip on the products they take over: a single 128-column operand (also as the second
    operand when no row carries the first), row-limited output, BatchNorm column sums, backward-data through the
    weight as stored; shapes they do not take (N not a multiple of 128) must fall through.
    bf16-pieces (stemgnn_linear_set_pair(0), csrc/wsgemm.hip): the tile kernel's BITS; two live operands fall through.
    pair (the default, csrc/wspair.hip): fp32-accurate results from two fp16 pieces per scaled row -- compared with the
    tile kernel at a few units of fp32 resolution of the row's product sum -- and the sampled batch's two-operand layer
    product too: the aggregate's rows (a boundary inside a tile, one row only, more head tiles than a quarter of the
    blocks) multiplied in the blocks' prologues."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(m + n)
    st = torch.cuda.current_stream().cuda_stream
    k = 128
    a = torch.randn(m, k, device=dev) * (1 + 3 * torch.rand(m, 1, device=dev))
    w = torch.randn(n, k, device=dev) * 0.2
    a2 = torch.randn(m, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.2 if k2 else None
    b = torch.randn(n, device=dev)
    if rows >= 0:
        a[rows:] = 0
    sr = store if store >= 0 else m

    def run():
        y = torch.full((m, n), 7.0, device=dev)
        part = torch.zeros(max(int(lib.stemgnn_linear_stats_blocks(m, n)), 1), 2, n, device=dev)
        check(lib.stemgnn_linear_fwd_rows(a.data_ptr(), w.data_ptr(), k, a2.data_ptr() if k2 else None,
                                          w2.data_ptr() if k2 else None, k2, b.data_ptr(), m, n, y.data_ptr(), part.data_ptr(),
                                          None, rows, sr, st))
        dy = torch.randn(m, n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        return y, part, ops.linear_bwd_data(dy, w) if n == 128 else None

    prev, was_pair = lib.stemgnn_linear_set_ws(0), ops.linear_set_pair(pair)
    try:
        y0, p0, d0 = run()
        lib.stemgnn_linear_set_ws(1)  # every eligible product, whatever its size
        calls = lib.stemgnn_linear_wsp_calls()
        y1, p1, d1 = run()
        calls = lib.stemgnn_linear_wsp_calls() - calls
    finally:
        lib.stemgnn_linear_set_ws(prev)
        ops.linear_set_pair(was_pair)
    assert bool((y1[sr:] == 7.0).all())
    ref = a.double() @ w.double().t() + b.double() + (a2.double() @ w2.double().t() if k2 else 0)
    if pair == 0:
        assert calls == 0
        assert torch.equal(y1, y0)
        if d0 is not None:
            assert torch.equal(d1, d0)
    else:
        # two live operands: taken when the aggregate's tiles are at most one per block (two blocks per CU and 128 columns)
        two_live = k2 > 0 and 0 < rows < m and n % 128 == 0 and (rows + 63) // 64 <= min(512 // (n // 128), (m + 63) // 64)
        fwd_taken = n % 128 == 0 and (k2 == 0 or rows == 0 or two_live)
        assert calls == int(fwd_taken) + int(d0 is not None), (calls, fwd_taken)
        # fp32 resolution of a row's product sum: |x| |w| summed over the contraction
        den = (a.abs() @ w.abs().t() + (a2.abs() @ w2.abs().t() if k2 else 0) + b.abs())[:sr]
        # (against fp64: 8 roundings of the row's product sum; against the tile kernel, which has its own: 12)
        assert float(((y1[:sr] - y0[:sr]).abs() / den).max()) < 12 * 2.0 ** -24
        assert float(((y1[:sr].double() - ref[:sr]).abs() / den.double()).max()) < 8 * 2.0 ** -24
        if d0 is not None:
            dy = torch.randn(m, n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
            dden = dy.abs() @ w.abs()
            assert float(((d1 - d0).abs() / dden).max()) < 8 * 2.0 ** -24
            assert float(((d1.double() - dy.double() @ w.double()).abs() / dden.double()).max()) < 8 * 2.0 ** -24
    torch.testing.assert_close(p1.sum(0), p0.sum(0), rtol=1e-6, atol=1e-4 * max(p0.sum(0).abs().max().item(), 1.0))
    torch.testing.assert_close(y1[:sr].double(), ref[:sr], rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(p1.sum(0)[0].double(), ref.sum(0), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(p1.sum(0)[1].double(), (ref * ref).sum(0), rtol=1e-4, atol=2e-2)


def test_pair_weight_stationary_product_with_non_finite_and_extreme_rows(dev):
    """Rows the power-of-two row factor has to survive (csrc/wspair.hip: pair_scale): a NaN, an Inf, a row of zeros, a
    row of denormal-sized values, a row near FLT_MAX and one huge element among tiny ones.  Non-finite rows come back
    non-finite in exactly those rows (as the tile kernel returns them), every other row is fp32-accurate -- the factor
    is clamped to 2^+-110, so the tiny row's product does not overflow and the huge row's does not vanish."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    torch.manual_seed(3)
    m, n, k = 20_000, 128, 128
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.1
    b = torch.randn(n, device=dev)
    x[5, 7] = float("nan")
    x[6, 9] = float("inf")
    x[7] = 0
    x[8] = torch.randn(k, device=dev) * 1e-39           # denormals
    x[9] = torch.randn(k, device=dev) * 1e-30
    x[10] = torch.randn(k, device=dev).clamp(-3, 3) * 1e36
    x[11] = torch.randn(k, device=dev) * 1e-12
    x[11, 3] = 5e5                                       # one element 2^58 above the rest of its row
    calls = lib.stemgnn_linear_wsp_calls()
    y = ops.linear_fwd(x, w, None, None, b, False)[0]
    assert lib.stemgnn_linear_wsp_calls() == calls + 1
    bad = ~torch.isfinite(y).all(dim=1)
    assert bad.nonzero().flatten().tolist() == [5, 6]
    ref = x.double() @ w.double().t() + b.double()
    good = ~bad
    den = (x.abs().double() @ w.abs().double().t() + b.abs().double())[good]
    assert float(((y[good].double() - ref[good]).abs() / den).max()) < 8 * 2.0 ** -24
    assert torch.equal(y[7], b)
    # the outlier row: the small elements are kept to 2^-39 of the row's largest (the format's second error term)
    err11 = (y[11].double() - ref[11]).abs()
    assert float(err11.max()) <= 8 * 2.0 ** -24 * float(den[11 - 2].max()) + 4 * 2.0 ** -39 * 5e5 * float(w.abs().sum(1).max())


@pytest.mark.parametrize("m,n,k", [(1024, 128, 128), (11000, 128, 256), (33, 96, 48), (1, 32, 16), (5000, 256, 128)])
def test_few_row_product_returns_the_tile_kernels_bits(dev, m, n, k):
    """stemgnn_linear_few_rows (one wave per 32 x 32 tile, operands straight from global memory) against the tile
    kernel and torch: forward with bias, and the backward-data form through the weight as stored."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(m + n + k)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(m, k, device=dev) * (1 + 3 * torch.rand(m, 1, device=dev))
    w = torch.randn(n, k, device=dev) * 0.2
    b = torch.randn(n, device=dev)
    prev = lib.stemgnn_linear_set_ws(0)
    try:
        y0 = ops.linear_fwd(x, w, None, None, b, False)[0]
        dy = torch.randn(m, n, device=dev)
        d0 = ops.linear_bwd_data(dy, w)
    finally:
        lib.stemgnn_linear_set_ws(prev)
    y1 = torch.full((m, n), 7.0, device=dev)
    check(lib.stemgnn_linear_few_rows(x.data_ptr(), w.data_ptr(), b.data_ptr(), m, n, k, y1.data_ptr(), 0, st))
    assert torch.equal(y1, y0)
    torch.testing.assert_close(y1, x @ w.t() + b, rtol=1e-4, atol=1e-3)
    if k % 32 == 0 and n % 16 == 0:
        d1 = torch.full((m, k), 7.0, device=dev)
        check(lib.stemgnn_linear_few_rows(dy.data_ptr(), w.data_ptr(), None, m, k, n, d1.data_ptr(), 1, st))
        assert torch.equal(d1, d0)


@pytest.mark.parametrize("pair", [0, 1], ids=["bf16-pieces", "pair"])
@pytest.mark.parametrize("N,H", [(20000, 4), (16390, 2)])
def test_vq_assign_weight_stationary_matches_the_tile_form(dev, N, H, pair):
    """The lean code assignment at K = Dc = 128 on the weight-stationary skeletons against k_vq_assign and against torch:
    indices, row norms, commitment sum; a zero row and a duplicated code (tie: lowest index) included.
    bf16-pieces (csrc/wsgemm.hip: k_vq_assign_ws): the tile form's arithmetic, the same indices.
    pair (the default, csrc/wspair.hip: k_vq_assign_wsp): an fp32-accurate product summed in another order -- an index may
    differ from the tile form's only where the row's two best similarities are within rounding of each other."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(N + H)
    st = torch.cuda.current_stream().cuda_stream
    K = Dc = 128
    xp = torch.randn(N, H * Dc, device=dev) * (0.5 + torch.rand(N, 1, device=dev))
    xp[7] = 0
    embed = torch.nn.functional.normalize(torch.randn(H, K, Dc, device=dev), dim=-1) * (1 + 0.05 * torch.rand(H, K, 1, device=dev))
    embed[:, 90] = embed[:, 17]  # a tie wherever code 17 wins
    embed = embed.contiguous()
    esq = (embed * embed).sum(-1).contiguous()

    def run():
        norm = torch.empty(N, H, device=dev)
        ind = torch.empty(N, H, dtype=torch.int64, device=dev)
        sq = torch.empty(1, device=dev)
        ws = torch.empty(int(lib.stemgnn_vq_workspace_bytes(N, H, Dc, K)), dtype=torch.uint8, device=dev)
        check(lib.stemgnn_vq_assign_lean(xp.data_ptr(), N, H, Dc, embed.data_ptr(), esq.data_ptr(), K, norm.data_ptr(),
                                         ind.data_ptr(), sq.data_ptr(), 0.25, ws.data_ptr(), ws.numel(), st))
        return norm, ind, sq, lib.stemgnn_vq_assign_last_path()

    prev, was_pair = lib.stemgnn_linear_set_ws(0), ops.linear_set_pair(pair)
    try:
        n0, i0, s0, p0 = run()
        lib.stemgnn_linear_set_ws(1)
        n1, i1, s1, p1 = run()
    finally:
        lib.stemgnn_linear_set_ws(prev)
        ops.linear_set_pair(was_pair)
    assert (p0, p1) == (1, 3 if pair else 2)
    xh = xp.view(N, H, Dc)
    if pair == 0:
        assert torch.equal(i1, i0)
    else:
        sim64 = torch.einsum("nhd,hkd->nhk", xh.double(), embed.double())
        top2 = sim64.topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1]) / xh.double().norm(dim=-1).clamp_min(1e-30)
        differ = i1 != i0
        assert float(differ.float().mean()) < 1e-3 and bool((gap[differ] < 1e-5).all())
        assert bool((i1[7] == i0[7]).all())  # the zero row: every similarity equal, index 0
    # the row norms: the same values summed in another order -- a few units in the last place
    rel = ((n1 - n0).abs() / n0.clamp_min(1e-30)).max().item()
    assert rel <= 4e-7, rel
    torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(n1, xh.norm(dim=-1), rtol=1e-5, atol=1e-6)
    sim = torch.einsum("nhd,hkd->nhk", torch.nn.functional.normalize(xh, dim=-1), embed)
    ref = sim.argmax(-1)
    agree = (ref == i1).float().mean().item()
    assert agree > 0.999  # fp32 rounding differs in the last bit between the two products; exact ties go to the lowest index
    assert int((i1 == 90).sum()) == 0 or bool(((i1 == 90) <= (ref != 17)).all())
    assert int((i1 == 90).sum()) == 0  # the duplicate never wins: the lowest index does


@pytest.mark.parametrize("N,H,K,Dc", [(9000, 2, 512, 256), (8192, 3, 2048, 768)])
def test_vq_assign_large_codebook_on_the_bigtile_core(dev, N, H, K, Dc):
    """The code assignment at large codebooks (K >= 512, Dc >= 256, N >= 8 192: BASELINE configs 3 / 5) on the big-tile
    core (csrc/bigtile.hip: bt_vq_assign -- the six exact piece products of all heads as one launch, the arg-max taken
    from the accumulators per 256-code tile, no [N, K] matrix) against the fused tile kernel (core switched off) and
    against torch: indices (lowest index among equals; a duplicated code and a zero row included), row norms, commitment
    sum.  The two sum their fp32 products in different orders: an index may differ only where the top-2 similarity gap
    is at rounding level."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    ops.linear_scratch(N, Dc, Dc, vq=(H, Dc, K))  # the core's scratch is the caller's
    torch.manual_seed(N + K)
    st = torch.cuda.current_stream().cuda_stream
    xp = torch.randn(N, H * Dc, device=dev) * (0.5 + torch.rand(N, 1, device=dev))
    xp[7] = 0
    embed = torch.nn.functional.normalize(torch.randn(H, K, Dc, device=dev), dim=-1) * (1 + 0.05 * torch.rand(H, K, 1, device=dev))
    embed[:, 90] = embed[:, 17]  # a tie wherever code 17 wins
    embed = embed.contiguous()
    esq = (embed * embed).sum(-1).contiguous()

    def run():
        norm = torch.empty(N, H, device=dev)
        ind = torch.empty(N, H, dtype=torch.int64, device=dev)
        sq = torch.empty(1, device=dev)
        ws = torch.empty(int(lib.stemgnn_vq_workspace_bytes(N, H, Dc, K)), dtype=torch.uint8, device=dev)
        check(lib.stemgnn_vq_assign_lean(xp.data_ptr(), N, H, Dc, embed.data_ptr(), esq.data_ptr(), K, norm.data_ptr(),
                                         ind.data_ptr(), sq.data_ptr(), 0.25, ws.data_ptr(), ws.numel(), st))
        return norm, ind, sq, lib.stemgnn_vq_assign_last_path()

    prev = lib.stemgnn_linear_set_bigtile(0)
    try:
        n0, i0, s0, p0 = run()
        lib.stemgnn_linear_set_bigtile(1)
        n1, i1, s1, p1 = run()
    finally:
        lib.stemgnn_linear_set_bigtile(prev)
    assert (p0, p1) == (1, 4)  # the tile kernel, then the big-tile core
    xh = xp.view(N, H, Dc)
    sim = torch.einsum("nhd,hkd->nhk", xh.double(), embed.double())
    top2 = sim.topk(2, dim=-1).values
    gap = ((top2[..., 0] - top2[..., 1]) / xh.double().norm(dim=-1).clamp_min(1e-30))
    differ = i1 != i0
    assert float(differ.float().mean()) < 1e-3 and bool((gap[differ] < 1e-5).all())
    assert int((i1 == 90).sum()) == 0  # the duplicate never wins: the lowest index does
    assert bool((i1[7] == i0[7]).all())  # the zero row: every similarity equal, index 0
    torch.testing.assert_close(n1, n0, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(n1, xh.norm(dim=-1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-6)

    # the form that hands the codes on (what the module-level path of the bf16 GEMM mode calls): straight-through rows,
    # normalised rows, the commitment sum from the gathered codes
    def run_full(training):
        xn, q = torch.empty_like(xp), torch.empty_like(xp)
        norm = torch.empty(N, H, device=dev)
        ind = torch.empty(N, H, dtype=torch.int64, device=dev)
        sq = torch.empty(1, device=dev)
        ws = torch.empty(int(lib.stemgnn_vq_workspace_bytes(N, H, Dc, K)), dtype=torch.uint8, device=dev)
        check(lib.stemgnn_vq_assign_fwd(xp.data_ptr(), N, H, Dc, embed.data_ptr(), K, training, xn.data_ptr(), norm.data_ptr(),
                                        ind.data_ptr(), q.data_ptr(), sq.data_ptr(), 0.25, ws.data_ptr(), ws.numel(), st))
        return xn, q, ind, sq, lib.stemgnn_vq_assign_last_path()

    for training in (1, 0):
        lib.stemgnn_linear_set_bigtile(0)
        try:
            xn0, q0, j0, t0, p0 = run_full(training)
            lib.stemgnn_linear_set_bigtile(1)
            xn1, q1, j1, t1, p1 = run_full(training)
        finally:
            lib.stemgnn_linear_set_bigtile(prev)
        assert (p0, p1) == (1, 4)
        same = (j1 == j0).unsqueeze(-1).expand(N, H, Dc).reshape(N, H * Dc)
        assert float((j1 != j0).float().mean()) < 1e-3
        torch.testing.assert_close(xn1, xn0, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(q1[same], q0[same], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(t1, t0, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n,e,d", [(300, 2000, 64), (1000, 9000, 128), (50, 0, 32), (40, 300, 256)])
def test_deterministic_decoder_scatters_match_the_atomic_ones(dev, n, e, d):
    """stemgnn_edge_dot_bwd_det / stemgnn_edge_concat_bwd_det (edges grouped by node, fixed order) against the atomic
    scatters and torch; two runs return the same bits."""
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(n + e)
    st = torch.cuda.current_stream().cuda_stream
    ei = torch.randint(0, n, (2, e), device=dev)
    z = torch.randn(n, d, device=dev)
    coef, gs = torch.randn(max(e, 1), device=dev), torch.tensor([0.7], device=dev)
    ws = torch.empty(int(lib.stemgnn_edge_det_workspace_bytes(n, e)), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        g = torch.full((n, d), 7.0, device=dev)
        check(lib.stemgnn_edge_dot_bwd_det(coef.data_ptr(), gs.data_ptr(), z.data_ptr(), n, d, ei.data_ptr(), e, g.data_ptr(),
                                           ws.data_ptr(), ws.numel(), st))
        outs.append(g)
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(n, d, device=dev)
    if e:
        w = (coef[:e] * gs)[:, None]
        ref.index_add_(0, ei[0], w * z[ei[1]])
        ref.index_add_(0, ei[1], w * z[ei[0]])
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=1e-4)
    go = torch.randn(max(e, 1), 2 * d, device=dev)
    base = torch.randn(n, d, device=dev)
    outs = []
    for _ in range(2):
        g = base.clone()
        check(lib.stemgnn_edge_concat_bwd_det(go.data_ptr(), n, d, ei.data_ptr(), e, g.data_ptr(), ws.data_ptr(), ws.numel(), st))
        outs.append(g)
    assert torch.equal(outs[0], outs[1])
    ref = base.clone()
    if e:
        ref.index_add_(0, ei[0], go[:e, :d])
        ref.index_add_(0, ei[1], go[:e, d:])
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=1e-4)


def test_linear_row_limited_output(dev):
    """stemgnn_linear_fwd_rows: rows past store_rows feed the column statistics but are not written."""
    from stem_gnn_amd._lib import lib, check
    from stem_gnn_amd import ops
    torch.manual_seed(0)
    m, k, n, r = 1500, 64, 128, 200
    x, w, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.randn(n, device=dev)
    y0, p0, blocks = ops.linear_fwd(x, w, None, None, b, True)
    y = torch.full((r + 50, n), 7.0, device=dev)
    part = torch.empty_like(p0)
    check(lib.stemgnn_linear_fwd_rows(x.data_ptr(), w.data_ptr(), k, None, None, 0, b.data_ptr(), m, n, y.data_ptr(),
                                      part.data_ptr(), None, -1, r, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(y[:r], y0[:r]) and bool((y[r:] == 7.0).all())
    assert torch.equal(part, p0)


def test_sage_agg_bwd_accumulates(dev):
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    from stem_gnn_amd.graph import GraphStructure
    n, e, d = 400, 3000, 64
    torch.manual_seed(1)
    ei = torch.stack([torch.randint(0, 300, (e,)), torch.randint(0, n, (e,))])  # nodes 300.. have no out-edges at all
    gs = GraphStructure(ei.to(dev), n).ensure_transpose()
    x, g, base = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev), torch.randn(n, d, device=dev)
    ref = ops.sage_agg_bwd(g, x, gs, None, None)
    out = base.clone()
    check(lib.stemgnn_sage_agg_bwd_acc(g.data_ptr(), x.data_ptr(), n, d, gs.rowptr_t.data_ptr(), gs.dst_t.data_ptr(),
                                       gs.eid_t.data_ptr(), gs.inv_deg.data_ptr(), None, None, None, 0, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
    torch.testing.assert_close(out, base + ref, rtol=1e-6, atol=1e-6)
    no_out = torch.bincount(ei[0], minlength=n) == 0
    assert bool(no_out.any()) and torch.equal(out[no_out.to(dev)], base[no_out.to(dev)])  # untouched, not zeroed


@pytest.mark.parametrize("N,H,K,D,Dc", [(1000, 4, 128, 128, 128), (333, 2, 40, 96, 48), (50, 1, 8, 32, 32), (2500, 4, 16, 64, 64),
                                        (1000, 4, 64, 128, 32), (700, 6, 24, 128, 32), (300, 5, 12, 64, 8),
                                        (9001, 4, 128, 128, 128), (20000, 2, 64, 128, 128)])
def test_vq_project_out_algebra_and_fused_backward(dev, N, H, K, D, Dc):
    """The code-table form of project_out (table read, segment-sum weight gradient) and the assignment backward with
    project_out's backward-data product inside, against the plain products they replace.  From 8 192 rows at D = Dc = 128
    the fused backward runs on the pair-format weight-stationary skeleton (csrc/wspair.hip, EPI = 1: a ragged last tile
    and a row under the eps clamp included)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(N + K)
    st = torch.cuda.current_stream().cuda_stream
    HD = H * Dc
    embed = torch.nn.functional.normalize(torch.randn(H, K, Dc, device=dev), dim=-1) * (1 + 0.1 * torch.rand(H, K, 1, device=dev))
    w_out, b_out = torch.randn(D, HD, device=dev) * 0.1, torch.randn(D, device=dev)
    ind = torch.randint(0, K, (N, H), device=dev)
    codes = torch.stack([embed[h][ind[:, h]] for h in range(H)], dim=1).reshape(N, HD)
    # forward: table + gather-sum == project_out(codes)
    table = torch.empty(H, K, D, device=dev)
    check(lib.stemgnn_small_gemm(embed.data_ptr(), Dc, 1, K * Dc, w_out.data_ptr(), 1, HD, Dc, table.data_ptr(), D, 1, K * D,
                                 K, D, Dc, H, st))
    out = torch.empty(N, D, device=dev)
    check(lib.stemgnn_codes_project(table.data_ptr(), ind.data_ptr(), b_out.data_ptr(), N, H, K, D, out.data_ptr(), st))
    torch.testing.assert_close(out, codes @ w_out.t() + b_out, rtol=1e-4, atol=1e-5)
    # backward: segment sums -> dW_out, db_out
    g = torch.randn(N, D, device=dev)
    sums = torch.empty(H * K, D, device=dev)
    ws = torch.empty(lib.stemgnn_code_segment_sums_workspace_bytes(N, H, K, D), dtype=torch.uint8, device=dev)
    check(lib.stemgnn_code_segment_sums(ind.data_ptr(), H, K, g.data_ptr(), N, D, sums.data_ptr(), ws.data_ptr(), ws.numel(), st))
    ref_sums = torch.zeros(H, K, D, device=dev)
    for h in range(H):
        ref_sums[h].index_add_(0, ind[:, h], g)
    torch.testing.assert_close(sums.view(H, K, D), ref_sums, rtol=1e-4, atol=1e-4)
    gw = torch.empty(D, HD, device=dev)
    check(lib.stemgnn_small_gemm(sums.data_ptr(), 1, D, K * D, embed.data_ptr(), Dc, 1, K * Dc, gw.data_ptr(), HD, 1, Dc,
                                 D, Dc, K, H, st))
    torch.testing.assert_close(gw, g.t() @ codes, rtol=1e-4, atol=1e-3)
    gb = torch.empty(D, device=dev)
    check(lib.stemgnn_segment_colsum(sums.data_ptr(), K, D, gb.data_ptr(), st))
    torch.testing.assert_close(gb, g.sum(0), rtol=1e-4, atol=1e-3)
    # fused assignment backward == backward-data product + assignment backward
    xp = torch.randn(N, HD, device=dev)
    xp[3] = 0  # a row under the eps clamp of F.normalize
    xp[5, :Dc] = 0  # ... and one with only its first head there (heads share a column tile when Dc divides 128)
    norm = xp.view(N, H, Dc).norm(dim=-1).contiguous()
    g_loss = torch.tensor([0.7], device=dev)
    g_q = ops.linear_bwd_data(g, w_out)
    ref = torch.empty_like(xp)
    check(lib.stemgnn_vq_assign_bwd(g_q.data_ptr(), g_loss.data_ptr(), 10.0, xp.data_ptr(), norm.data_ptr(), ind.data_ptr(),
                                    embed.data_ptr(), N, H, Dc, K, ref.data_ptr(), st))
    got = torch.full_like(xp, float("nan"))
    calls = lib.stemgnn_linear_wsp_calls()
    check(lib.stemgnn_vq_assign_bwd_fused(g.data_ptr(), D, w_out.data_ptr(), g_loss.data_ptr(), 10.0, xp.data_ptr(),
                                          norm.data_ptr(), ind.data_ptr(), embed.data_ptr(), N, H, Dc, K, got.data_ptr(), st))
    assert lib.stemgnn_linear_wsp_calls() - calls == int(N >= 8192 and D == 128 and Dc == 128)
    scale = ref.abs().max().item()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-6 * max(scale, 1.0))
    if N >= 8192:  # ... and against the bf16-piece tile form of the same fused kernel
        was = ops.linear_set_pair(0)
        try:
            old = torch.full_like(xp, float("nan"))
            check(lib.stemgnn_vq_assign_bwd_fused(g.data_ptr(), D, w_out.data_ptr(), g_loss.data_ptr(), 10.0, xp.data_ptr(),
                                                  norm.data_ptr(), ind.data_ptr(), embed.data_ptr(), N, H, Dc, K, old.data_ptr(), st))
        finally:
            ops.linear_set_pair(was)
        torch.testing.assert_close(got, old, rtol=1e-4, atol=1e-6 * max(scale, 1.0))


# ---------------------------------------------------------------------------- round-3: the "last block finishes" protocol
def _fixed_order_finish(partials: np.ndarray, mul: float) -> np.float32:
    """finish_sum_block (csrc/loss_ops.hip): thread t adds partial[t], partial[t + 256], ... in fp64, then a pairwise
    tree over the 256 threads; the product with `mul` is rounded to fp32 once."""
    n = partials.size
    pad = np.zeros((n + 255) // 256 * 256, dtype=np.float64)
    pad[:n] = partials
    lanes = np.zeros(256, dtype=np.float64)
    for row in pad.reshape(-1, 256):  # sequential per thread
        lanes = lanes + row
    o = 128
    while o > 0:
        lanes[:o] = lanes[:o] + lanes[o:2 * o]
        o //= 2
    return np.float32(lanes[0] * mul)


def test_ticketed_reduction_many_blocks_many_rounds(dev):
    """The fence-free last-block hand-off (csrc/common.h: st_agent / wait_stores / ticket_last / ld_agent) is an observed
    property of gfx950, not a guarantee of the HIP memory model.  Pin it: a reduction whose 4 096
    blocks cover every XCD (16 rounds of resident blocks, so late blocks start long after early ones finished), 150
    rounds in one process with fresh inputs each round (a partial read stale from the round before changes the sum),
    each result compared BIT FOR BIT with the fixed-order sum of the partials the kernel itself wrote -- the value
    the two-launch k_finish_sum path returns."""
    from stem_gnn_amd._lib import lib, check
    rows, D = 16384, 256
    st = torch.cuda.current_stream().cuda_stream
    ws_bytes = int(lib.stemgnn_loss_workspace_bytes(rows))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
    row_loss = ws[base:base + rows * 8].view(torch.float64)
    save = torch.empty(rows, 3, device=dev)
    loss = torch.empty(1, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    bad = 0
    for it in range(150):
        z = torch.randn(rows, D, device=dev, generator=g)
        h = torch.randn(rows, D, device=dev, generator=g)
        check(lib.stemgnn_cosine_loss_fwd(z.data_ptr(), h.data_ptr(), rows, D, 1.0, loss.data_ptr(), save.data_ptr(),
                                          ws.data_ptr(), ws_bytes, st), "cosine_loss_fwd")
        got = loss.cpu().numpy()[0]
        exp = _fixed_order_finish(row_loss.cpu().numpy(), 1.0 / rows)
        bad += int(got.tobytes() != exp.tobytes())
        # the partials themselves are the row losses 1 - cos (fp32 arithmetic, widened)
        torch.testing.assert_close(row_loss.float(), 1.0 - save[:, 0], rtol=0, atol=0)
    assert bad == 0, f"{bad} of 150 ticketed sums differ from the fixed-order sum of their own partials"


def test_two_streams_reduce_side_by_side(dev):
    """Round 2 hashed the OUTPUT POINTER of a reduction into a pool of 61 counter words, so two reductions in flight on
    two streams whose outputs collided mod 61 corrupted each other's "last block" decision.
    Counters now belong to the (device, stream) pair.  Two streams run mean-squared-error reductions at the same time,
    200 rounds, into outputs that are 61 floats apart (the old colliding keys); both must return their own sums."""
    from stem_gnn_amd._lib import lib, check
    n = 1 << 20
    out = torch.zeros(62, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ws_bytes = int(lib.stemgnn_loss_workspace_bytes(256))
    ws1 = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ws2 = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a = torch.randn(4, n, device=dev)
    b = torch.randn(4, n, device=dev)
    torch.cuda.synchronize()
    exp1 = [float(((a[i] - b[i]).double() ** 2).mean()) for i in range(4)]
    exp2 = [float(((a[i] - 2 * b[i]).double() ** 2).mean()) for i in range(4)]
    b2 = 2 * b
    torch.cuda.synchronize()
    for it in range(200):
        i = it % 4
        check(lib.stemgnn_mse_loss_fwd(a[i].data_ptr(), b[i].data_ptr(), n, 1.0, out[0:].data_ptr(), ws1.data_ptr(),
                                       ws_bytes, s1.cuda_stream), "mse s1")
        check(lib.stemgnn_mse_loss_fwd(a[i].data_ptr(), b2[i].data_ptr(), n, 1.0, out[61:].data_ptr(), ws2.data_ptr(),
                                       ws_bytes, s2.cuda_stream), "mse s2")
        s1.synchronize()
        s2.synchronize()
        r = out.cpu()
        assert abs(float(r[0]) - exp1[i]) <= 1e-5 * exp1[i], (it, float(r[0]), exp1[i])
        assert abs(float(r[61]) - exp2[i]) <= 1e-5 * exp2[i], (it, float(r[61]), exp2[i])
        out.zero_()
        torch.cuda.synchronize()
