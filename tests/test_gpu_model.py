"""GPU parity tests at module level (run with -m gpu): VectorQuantize against the reference's
golden vectors (tests/golden/vq_*.pt, produced by importing the reference's model/vq.py) and
against the CPU oracle at larger sizes; Encoder and the full pretraining step against the CPU
oracle with the random draws of the HIP run replayed."""
import glob
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import stem_oracle as O  # noqa: E402  (checker only)

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vq_*.pt")))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def assert_indices_match(ind, ref, gap, tol=1e-5):
    """Bit-exact wherever the reference's top-2 similarity gap exceeds `tol`; a near-tie may
    resolve either way because the fp32 summation order of the MFMA chain differs from ATen's
    GEMM (SURVEY §7 'VQ arg-max exactness')."""
    ind, ref, gap = ind.reshape(-1), ref.reshape(-1), gap.reshape(-1)
    diff = ind != ref
    assert not bool((diff & (gap > tol)).any()), f"{int((diff & (gap > tol)).sum())} index mismatches beyond ties"
    return int(diff.sum())


def build_vq(N, D, H, K, Dc, ortho_max, ema, dev):
    from stem_gnn_amd.model.vq import VectorQuantize
    return VectorQuantize(dim=D, codebook_size=K, codebook_dim=Dc, heads=H, separate_codebook_per_head=True,
                          decay=0.8, commitment_weight=10, use_cosine_sim=True, orthogonal_reg_weight=1,
                          orthogonal_reg_max_codes=ortho_max, orthogonal_reg_active_codes_only=False,
                          kmeans_init=False, ema_update=bool(ema)).to(dev)


def _rows_equal(got, want, keep_rows, rtol, atol, what):
    """assert_close on the rows flagged in keep_rows (all of them unless an index flipped)."""
    assert int(keep_rows.sum()) >= 0.99 * keep_rows.numel(), f"{what}: more than 1 % of the rows carry a flipped index"
    torch.testing.assert_close(got[keep_rows], want[keep_rows], rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


@pytest.mark.parametrize("lean", [False, True], ids=["codes", "phase"])
@pytest.mark.parametrize("gemm_mode", [1, 0], ids=["bf16x3", "f32mfma"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_vq_matches_reference_golden(dev, path, gemm_mode, lean, record_property):
    """Every value the reference produced for the fixture is compared on every row whose code indices agree -- a
    near-tie flip (top-2 gap < 1e-5) removes only its own row from the row-wise comparisons, never the whole
    fixture; sums over rows (loss, parameter gradients, EMA buffers) are compared exactly when no index flipped and
    with the flipped rows' bounded contribution otherwise.  Both matrix-core paths (exact bf16 pieces / fp32 MFMA)
    run every fixture, and so does the one-call module phase (``skip_codes``: project_out read off the projected code
    table, project_out's weight gradient from per-code segment sums; it returns no per-head codes and leaves the EMA
    codebook update to the per-op path)."""
    from stem_gnn_amd import ops
    fx = torch.load(path, weights_only=True)
    N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
    if lean and (ema or H * Dc == D):
        pytest.skip("the phase covers the gradient-trained codebook with projections (pretrain.py:104-119)")
    prev = ops.linear_set_mode(gemm_mode)
    try:
        vq = build_vq(N, D, H, K, Dc, ortho_max, ema, dev)
        vq.skip_codes = lean
        state = {k[len("state0."):]: v for k, v in fx.items() if k.startswith("state0.")}
        vq.load_state_dict(state)  # the reference's exact key / shape contract
        vq.train()
        z = fx["z"].to(dev).requires_grad_(True)
        # replay the reference's randperm draw for the orthogonal loss
        if K > ortho_max:
            vq._rand_code_ids = lambda n, k, device: fx["ortho_ids"].to(device)
        q, ind, loss, oq = vq(z)
        assert (oq is None) == lean
        assert ind.dtype == torch.int64 and tuple(ind.shape) == tuple(fx["train.embed_ind"].shape)
        flips = assert_indices_match(ind.cpu(), fx["train.embed_ind"], fx["top2_gap"])
        record_property("train_index_flips", flips)
        assert flips <= 0.01 * N * H, f"{flips} near-tie flips of {N * H} assignments"
        same = (ind.cpu().reshape(N, -1) == fx["train.embed_ind"].reshape(N, -1))          # [N, H]
        row_ok = same.all(dim=1)
        head_ok = same.repeat_interleave(Dc, dim=1)                                          # [N, H*Dc]
        if not lean:
            got_oq, want_oq = oq.detach().cpu(), fx["train.orig_quantize"]
            torch.testing.assert_close(torch.where(head_ok, got_oq, want_oq), want_oq, rtol=1e-4, atol=1e-5)
        _rows_equal(q.detach().cpu(), fx["train.quantize"], row_ok, 1e-4, 1e-5, "quantize")
        # commitment term: a flipped assignment moves one row-head's squared error by < 2 * gap (unit vectors)
        slack = 10.0 * 2.0 * 1e-5 * flips / max(N * H * Dc, 1)
        torch.testing.assert_close(loss.detach().cpu(), fx["train.loss"], rtol=1e-4, atol=1e-5 + slack)
        w = torch.linspace(-1.0, 1.0, q.numel()).view_as(q).to(dev)
        (loss.sum() + (q * w).sum()).backward()
        _rows_equal(z.grad.cpu(), fx["train.grad_z"], row_ok, 1e-3, 1e-5, "grad_z")
        if flips == 0:  # sums over all rows: a flipped row swaps a whole code vector in them
            for pn, p in vq.named_parameters():
                key = "train.grad." + pn
                if key in fx:
                    torch.testing.assert_close(p.grad.cpu(), fx[key], rtol=1e-3, atol=1e-5)
            if ema:
                for k, v in vq.state_dict().items():
                    if k.startswith("_codebook."):
                        torch.testing.assert_close(v.cpu(), fx["post." + k], rtol=1e-4, atol=1e-5)
        # eval mode on the initial state
        vq2 = build_vq(N, D, H, K, Dc, ortho_max, ema, dev)
        vq2.load_state_dict(state)
        vq2.skip_codes = lean
        vq2.eval()
        with torch.no_grad():
            q2, ind2, loss2, oq2 = vq2(fx["z"].to(dev))
        flips2 = assert_indices_match(ind2.cpu(), fx["eval.embed_ind"], fx["top2_gap"])
        record_property("eval_index_flips", flips2)
        assert flips2 <= 0.01 * N * H
        same2 = (ind2.cpu().reshape(N, -1) == fx["eval.embed_ind"].reshape(N, -1))
        _rows_equal(q2.cpu(), fx["eval.quantize"], same2.all(dim=1), 1e-4, 1e-5, "eval quantize")
        if not lean:
            head_ok2 = same2.repeat_interleave(Dc, dim=1)
            torch.testing.assert_close(torch.where(head_ok2, oq2.cpu(), fx["eval.orig_quantize"]),
                                       fx["eval.orig_quantize"], rtol=1e-4, atol=1e-5)
        assert float(loss2) == 0.0
    finally:
        ops.linear_set_mode(prev)


def test_golden_flip_budget(dev):
    """The golden comparisons above are not vacuous: the row-sum quantities (parameter gradients, EMA buffers) are
    only compared for fixtures without an index flip, so nearly all fixtures must be flip-free in each mode.  Counts
    the flips itself (one train-mode forward per fixture and mode), so it runs alone or under -k as well."""
    from stem_gnn_amd import ops
    for mode in (0, 1):
        prev = ops.linear_set_mode(mode)
        try:
            flipped = {}
            for path in FIXTURES:
                fx = torch.load(path, weights_only=True)
                N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
                vq = build_vq(N, D, H, K, Dc, ortho_max, ema, dev)
                vq.load_state_dict({k[len("state0."):]: v for k, v in fx.items() if k.startswith("state0.")})
                vq.train()
                if K > ortho_max:
                    vq._rand_code_ids = lambda n, k, device, ids=fx["ortho_ids"]: ids.to(device)
                _, ind, _, _ = vq(fx["z"].to(dev))
                n = assert_indices_match(ind.cpu(), fx["train.embed_ind"], fx["top2_gap"])
                if n:
                    flipped[os.path.basename(path)] = n
        finally:
            ops.linear_set_mode(prev)
        print(f"gemm mode {mode}: {len(FIXTURES)} fixtures, flips: {flipped or 'none'}")
        assert len(flipped) <= max(1, len(FIXTURES) // 6), flipped


import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import prod_recipe as R  # noqa: E402

PROD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "prod_vq_*.pt")))


@pytest.mark.parametrize("lean", [False, True], ids=["codes", "phase"])
@pytest.mark.parametrize("gemm_mode", [1, 0], ids=["bf16x3", "f32mfma"])
@pytest.mark.parametrize("path", PROD, ids=[os.path.basename(p) for p in PROD])
def test_vq_matches_reference_golden_at_production_shapes(dev, path, gemm_mode, lean):
    """reference-generated vectors at the shapes the production kernels run -- SURVEY.md
    section 8(c)'s (N=1000, D=128, H=4, K=512, Dc=128) and one case past the row gate of the weight-stationary
    assignment (N=16 500, K=Dc=128), which must actually be SERVED by k_vq_assign_ws (library query).  Inputs and state
    are regenerated from the fixture's seeds (tests/golden/prod_recipe.py, checksummed); the reference's results are
    compared through indices (bit-exact outside near-ties), the loss, and three checks per row (sum, fixed dot
    product, L1 norm) of quantize, grad_z and every parameter gradient: 1e-4 of the row's L1 norm."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    fx = torch.load(path, weights_only=True)
    N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
    state = R.make_state(D, H, K, Dc, seed)
    z_cpu = R.make_input(N, D, seed)
    for k, v in state.items():
        torch.testing.assert_close(R.checksum(v), fx["chk.state0." + k], rtol=1e-12, atol=1e-9, msg=f"recipe drift: {k}")
    torch.testing.assert_close(R.checksum(z_cpu), fx["chk.z"], rtol=1e-12, atol=1e-9, msg="recipe drift: z")
    want_ind, gap = fx["train.embed_ind"].long(), fx["top2_gap"].float()
    prev = ops.linear_set_mode(gemm_mode)
    try:
        vq = build_vq(N, D, H, K, Dc, ortho_max, 0, dev)
        vq.skip_codes = lean
        vq.load_state_dict(state)
        vq.train()
        vq._rand_code_ids = lambda n, k, device: fx["ortho_ids"].to(device)
        z = z_cpu.to(dev).requires_grad_(True)
        q, ind, loss, oq = vq(z)
        served_by = int(lib.stemgnn_vq_assign_last_path())
        if lean and gemm_mode == 1 and K == 128 and Dc == 128 and N >= 16384:
            # k_vq_assign_wsp (pair format, the default) / k_vq_assign_ws (stemgnn_linear_set_pair(0))
            assert served_by == (3 if ops.linear_set_pair(-1) else 2), \
                "a weight-stationary assignment kernel must serve the production-shape case"
        elif gemm_mode == 1 and K >= 512 and Dc >= 256 and N >= 8192:
            assert served_by == 4, "the big-tile core must serve the (9000, 256, 2, 512, 256) case"
        elif lean:
            assert served_by == 1
        flips = assert_indices_match(ind.cpu(), want_ind, gap)
        assert flips <= 20, f"{flips} near-tie flips of {N * H} assignments"
        same = ind.cpu().reshape(N, -1) == want_ind.reshape(N, -1)
        row_ok = same.all(dim=1)
        R.assert_rows_close(q.detach().cpu(), fx["train.quantize.rows"], row_ok, 1e-4, "quantize")
        if not lean:
            R.assert_rows_close(oq.detach().cpu(), fx["train.orig_quantize.rows"], row_ok, 1e-4, "orig_quantize")
        slack = 10.0 * 2.0 * 1e-5 * flips / max(N * H * Dc, 1)
        torch.testing.assert_close(loss.detach().cpu(), fx["train.loss"], rtol=1e-4, atol=1e-5 + slack)
        (loss.sum() + (q * R.make_upstream(N, D).to(dev)).sum()).backward()
        R.assert_rows_close(z.grad.cpu(), fx["train.grad_z.rows"], row_ok, 1e-3, "grad_z")
        if flips == 0:  # sums over all rows: a flipped row swaps a whole code vector in them
            for pn, p in vq.named_parameters():
                key = "train.grad." + pn + ".rows"
                if key in fx:
                    g = p.grad.cpu()
                    R.assert_rows_close(g if g.dim() > 1 else g.view(1, -1), fx[key], None, 1e-3, pn)
        if "eval.quantize.rows" in fx:
            vq.eval()
            with torch.no_grad():
                q2, ind2, loss2, _ = vq(z_cpu.to(dev))
            assert_indices_match(ind2.cpu(), want_ind, gap)
            ok2 = (ind2.cpu().reshape(N, -1) == want_ind.reshape(N, -1)).all(dim=1)
            R.assert_rows_close(q2.cpu(), fx["eval.quantize.rows"], ok2, 1e-4, "eval quantize")
            assert float(loss2) == 0.0
    finally:
        ops.linear_set_mode(prev)


@pytest.mark.parametrize("N,D,H,K,Dc", [(1, 32, 2, 8, 16), (127, 32, 4, 33, 32), (129, 64, 4, 128, 64),
                                        (1000, 128, 4, 512, 128), (300, 96, 2, 200, 100), (260, 768, 4, 128, 768),
                                        (9001, 128, 4, 128, 128)])
@pytest.mark.parametrize("lean", [False, True], ids=["codes", "phase"])
def test_vq_vs_oracle_sizes(dev, N, D, H, K, Dc, lean):
    """Ragged tiles (N not a multiple of 128, K not a multiple of 32, Dc not a multiple of 32), through the per-op
    path and through the one-call module phase.  (9001, 128, 4, 128, 128): past the row gate of the pair-format
    weight-stationary kernels -- in the phase form the backward runs the quantiser's fused backward and project_in's
    backward-data product (k_linear_ksp, rows scaled by the maxima the fused kernel wrote) from csrc/wspair.hip."""
    torch.manual_seed(N + K)
    ovq = O.OracleVectorQuantize(D, K, Dc, H, commitment_weight=10.0, orthogonal_reg_weight=1.0,
                                 orthogonal_reg_max_codes=32, ema_update=False)
    vq = build_vq(N, D, H, K, Dc, 32, False, dev)
    vq.load_state_dict(ovq.state_dict())
    vq.skip_codes = lean
    z = torch.randn(N, D)
    ids = torch.randperm(K)[:32] if K > 32 else None
    ovq.train(); vq.train()
    zr = z.clone().requires_grad_(True)
    qr, ir, lr, oqr = ovq(zr, ortho_ids=ids)
    with torch.no_grad():
        x = torch.nn.functional.normalize(ovq.project_in(z).view(N, H, Dc), dim=-1)
        sim = torch.einsum("nhd,hcd->nhc", x, ovq._codebook.embed)
        top2 = sim.topk(2, dim=-1).values
        gap = top2[..., 0] - top2[..., 1]
    zg = z.to(dev).requires_grad_(True)
    if ids is not None:
        vq._rand_code_ids = lambda n, k, device: ids.to(device)
    qg, ig, lg, oqg = vq(zg)
    flips = assert_indices_match(ig.cpu(), ir, gap)
    assert flips <= max(1, 0.01 * N * H), flips
    row_ok = (ig.cpu().reshape(N, -1) == ir.reshape(N, -1)).all(dim=1)
    torch.testing.assert_close(qg.detach().cpu()[row_ok], qr.detach()[row_ok], rtol=1e-4, atol=1e-5)
    slack = 10.0 * 2.0 * 1e-5 * flips / max(N * H * Dc, 1)
    torch.testing.assert_close(lg.detach().cpu(), lr.detach(), rtol=1e-4, atol=1e-5 + slack)
    w = torch.randn(N, D)
    (lr.sum() + (qr * w).sum()).backward()
    (lg.sum() + (qg * w.to(dev)).sum()).backward()
    torch.testing.assert_close(zg.grad.cpu()[row_ok], zr.grad[row_ok], rtol=1e-3, atol=1e-5)
    if flips == 0:
        for (n1, p1), (n2, p2) in zip(ovq.named_parameters(), vq.named_parameters()):
            assert n1 == n2
            scale = p1.grad.abs().max().item()
            torch.testing.assert_close(p2.grad.cpu(), p1.grad, rtol=1e-3, atol=1e-5 * max(scale, 1.0),
                                       msg=lambda m: f"{n1}: {m}")
    if N >= 8192 and lean:
        # the pair-format kernels of the phase (fused backward with row maxima, k_linear_ksp, the assignment) against the
        # bf16-piece kernels on the same call: the gradient of the input within a few fp32 roundings of the row's sums
        from stem_gnn_amd import ops
        from stem_gnn_amd._lib import lib
        calls = lib.stemgnn_linear_wsp_calls()
        z1 = z.to(dev).requires_grad_(True)
        q1, i1, l1, _ = vq(z1)
        (l1.sum() + (q1 * w.to(dev)).sum()).backward()
        # the fused backward and project_in's backward-data product (from 8 192 rows; project_in itself and the
        # assignment join them at 16 384: covered by the N = 16 500 golden case and by test_gpu_kernels.py)
        assert lib.stemgnn_linear_wsp_calls() - calls >= 2
        was = ops.linear_set_pair(0)
        try:
            calls = lib.stemgnn_linear_wsp_calls()
            z0 = z.to(dev).requires_grad_(True)
            q0, i0, l0, _ = vq(z0)
            (l0.sum() + (q0 * w.to(dev)).sum()).backward()
            assert lib.stemgnn_linear_wsp_calls() == calls
        finally:
            ops.linear_set_pair(was)
        same = (i1 == i0).all(dim=-1) if i1.dim() > 1 else (i1 == i0)
        assert float((~same).float().mean()) < 1e-3
        g1, g0 = z1.grad[same], z0.grad[same]
        row_l1 = g0.abs().sum(dim=1, keepdim=True).clamp_min(1e-30)
        assert float(((g1 - g0).abs() / row_l1).max()) < 2e-6
        torch.testing.assert_close(l1, l0, rtol=1e-5, atol=1e-6)


def test_vq_tie_breaks_to_lowest_index(dev):
    """Duplicate code rows: the arg-max must return the lowest index (torch.argmax semantics)."""
    from stem_gnn_amd import ops
    H, K, Dc, N = 2, 70, 32, 200
    torch.manual_seed(1)
    embed = torch.nn.functional.normalize(torch.randn(H, K, Dc), dim=-1)
    embed[:, 40] = embed[:, 3]    # same half of the MFMA tile
    embed[:, 69] = embed[:, 5]    # different code tile
    embed[:, 36] = embed[:, 33]   # other lane half (rows 4..7 live in the upper half)
    xp = torch.randn(N, H * Dc)
    xp[:50, :Dc] = embed[0, 3] * 2.0
    xp[50:100, :Dc] = embed[0, 5] * 0.5
    xp[100:150, Dc:] = embed[1, 33] * 3.0
    quant, ind, mse = ops.VqAssignFn.apply(xp.to(dev), embed.to(dev), H, True)
    ind = ind.cpu()
    assert (ind[:50, 0] == 3).all() and (ind[50:100, 0] == 5).all() and (ind[100:150, 1] == 33).all()


def make_models(D, L, H, K, Dc, dev, dropout=0.15, seed=0):
    from stem_gnn_amd.model.encoder import Encoder, InnerProductDecoder
    from stem_gnn_amd.model.pt_model import PretrainModel
    from stem_gnn_amd.model.vq import VectorQuantize
    torch.manual_seed(seed)
    om = O.build_oracle_model(D, L, H, K, Dc, dropout=dropout)
    enc = Encoder(D, D, nn.ReLU, L, backbone="sage", normalize="batch", dropout=dropout)
    vq = VectorQuantize(dim=D, codebook_size=K, codebook_dim=Dc, heads=H, separate_codebook_per_head=True, decay=0.8,
                        commitment_weight=10, use_cosine_sim=True, orthogonal_reg_weight=1,
                        orthogonal_reg_max_codes=32, kmeans_init=False, ema_update=False)
    gm = PretrainModel(enc, vq, nn.Linear(D, D), InnerProductDecoder(D, D), nn.Linear(2 * D, D))
    gm.load_state_dict(om.state_dict())  # identical parameter names/shapes by construction
    return om, gm.to(dev)


@pytest.mark.parametrize("attr", ["dense", "table", "none"])
def test_encoder_fwd_bwd_vs_oracle(dev, attr):
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    N, E, D, L = 500, 4000, 64, 3
    om, gm = make_models(D, L, 4, 64, D, dev)
    torch.manual_seed(5)
    x = torch.randn(N, D)
    ei = torch.randint(0, N, (2, E))
    table, et = torch.randn(6, D), torch.randint(0, 6, (E,))
    ea_cpu = None if attr == "none" else table[et]
    ea_gpu = None if attr == "none" else (ea_cpu.to(dev) if attr == "dense" else EdgeTypeAttr(table.to(dev), et.to(dev)))
    oe, ge = om.encoder, gm.encoder
    oe.train(); ge.train()
    xg = x.to(dev).requires_grad_(True)
    zg = ge(xg, ei.to(dev), ea_gpu)
    masks = [ops.dropout_keep_mask(N * D, 0.15, s, o, dev).view(N, D).cpu() for (s, o) in ge.last_dropout_keys]
    assert len(masks) == L - 1
    xr = x.clone().requires_grad_(True)
    zr = oe(xr, ei, ea_cpu, dropout_masks=masks)
    torch.testing.assert_close(zg.detach().cpu(), zr.detach(), rtol=1e-4, atol=1e-4)
    w = torch.randn(N, D)
    (zr * w).sum().backward()
    (zg * w.to(dev)).sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)
    for (n1, p1), (n2, p2) in zip(oe.named_parameters(), ge.named_parameters()):
        assert n1 == n2
        torch.testing.assert_close(p2.grad.cpu(), p1.grad, rtol=1e-3, atol=2e-4, msg=lambda m: f"{n1}: {m}")
    for (n1, b1), (n2, b2) in zip(oe.named_buffers(), ge.named_buffers()):
        assert n1 == n2
        torch.testing.assert_close(b2.cpu().float(), b1.float(), rtol=1e-4, atol=1e-5)
    # eval mode
    oe.eval(); ge.eval()
    with torch.no_grad():
        torch.testing.assert_close(ge(x.to(dev), ei.to(dev), ea_gpu).cpu(), oe(x, ei, ea_cpu), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N", [2048 * 3 + 77, 78_336 + 40])
def test_encoder_batchnorm_statistics_at_many_row_tiles(dev, N):
    """BatchNorm statistics gathered from the layer product's column-sum slabs at sizes with many row tiles and, at
    78 376 rows, a second launch of 32-row tail tiles writing into the same slab array (csrc/linear.hip plan_fwd).
    Output, running statistics and the batch counter against the CPU oracle, over two consecutive forwards."""
    from stem_gnn_amd import ops
    D, L, E = 128, 2, 3 * N
    om, gm = make_models(D, L, 4, 64, D, dev)
    torch.manual_seed(11)
    x = torch.randn(N, D) * (1 + torch.arange(D) / 32.0) + torch.arange(D) / 64.0  # per-column scale and offset
    ei = torch.randint(0, N, (2, E))
    oe, ge = om.encoder, gm.encoder
    oe.train(); ge.train()
    for rep in range(2):
        with torch.no_grad():
            zg = ge(x.to(dev), ei.to(dev), None)
            masks = [ops.dropout_keep_mask(N * D, 0.15, s, o, dev).view(N, D).cpu() for (s, o) in ge.last_dropout_keys]
            zr = oe(x, ei, None, dropout_masks=masks)
        torch.testing.assert_close(zg.cpu(), zr, rtol=1e-4, atol=1e-4)
        for (n1, b1), (n2, b2) in zip(oe.named_buffers(), ge.named_buffers()):
            assert n1 == n2
            torch.testing.assert_close(b2.cpu().float(), b1.float(), rtol=1e-5, atol=1e-6, msg=lambda m: f"{n1}: {m}")


@pytest.mark.parametrize("D,H,K,attr", [(768, 4, 128, "table"), (96, 2, 40, "dense"), (128, 4, 512, "table"),
                                        (256, 4, 2048, "table")])
def test_pretrain_step_other_widths(dev, D, H, K, attr):
    """One full step at the reference's default width (config/pretrain.yaml: D = Dc = 768, H = 4,
    K = 128; Cora-like size), at a non-power-of-two width with the dense edge_attr API, and at the codebook
    sizes of BASELINE configs 3 and 5 (K = 512, K = 2048)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import pretrain_step, default_params
    N, E, bs = 500, 3000, 128
    om, gm = make_models(D, 2, H, K, D, dev)
    params = default_params()
    torch.manual_seed(5)
    x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1)
    half = torch.randint(0, N, (2, E // 2))
    ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
    table = torch.nn.functional.normalize(torch.randn(5, D), dim=-1)
    et = torch.randint(0, 5, (E,))
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-4, weight_decay=1e-5)
    ea_g = EdgeTypeAttr(table.to(dev), et.to(dev)) if attr == "table" else table[et].to(dev)
    ops.manual_seed(7)
    loss_g, losses_g, draws = pretrain_step(gm, opt_g, None, params, x.to(dev), ei.to(dev), ea_g, bs)
    cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
    loss_o, losses_o, _ = O.pretrain_step(om, opt_o, None, params, x, ei, table[et], bs, cpu_draws)
    for k in losses_o:
        torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=1e-4, atol=1e-5,
                                   msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(loss_g.cpu().reshape(-1), loss_o.reshape(-1), rtol=1e-4, atol=1e-5)


def test_pretrain_steps_loss_parity(dev):
    """Several full pretraining steps (augment -> forward -> backward -> clip -> AdamW ->
    scheduler -> EMA teacher): the HIP path's loss curve against the CPU oracle replaying the
    HIP run's random draws.  Tolerance 1e-4 relative on every loss term (north_star)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import pretrain_step, default_params
    from stem_gnn_amd.utils.others import get_scheduler
    N, E, D, L, H, K = 600, 5000, 64, 2, 4, 64
    bs = 200
    om, gm = make_models(D, L, H, K, D, dev)
    params = default_params()  # the reference's lr 1e-4 (config/pretrain.yaml:18).  Adam turns rounding-level gradient
    # differences on near-zero-gradient elements into +-lr steps, so at a 10x larger lr a single such
    # element shows up as a transient 1e-4 loss difference (tests/parity_probe.py prints both regimes).
    torch.manual_seed(11)
    x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1)
    half = torch.randint(0, N, (2, E // 2))
    ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
    table = torch.nn.functional.normalize(torch.randn(4, D), dim=-1)
    et = torch.randint(0, 4, (E,))
    opt_o = torch.optim.AdamW(om.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    opt_g = torch.optim.AdamW(gm.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    sch_o, sch_g = get_scheduler(opt_o, True, 50), get_scheduler(opt_g, True, 50)
    xg, eig = x.to(dev), ei.to(dev)
    eag = EdgeTypeAttr(table.to(dev), et.to(dev))
    ops.manual_seed(99)
    for step in range(6):
        loss_g, losses_g, draws = pretrain_step(gm, opt_g, sch_g, params, xg, eig, eag, bs)
        cpu_draws = {}
        for k, v in draws.items():
            cpu_draws[k] = [m.cpu() for m in v] if isinstance(v, list) else v.cpu()
        loss_o, losses_o, ind_o = O.pretrain_step(om, opt_o, sch_o, params, x, ei, table[et], bs, cpu_draws)
        for k in losses_o:
            torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=1e-4, atol=1e-5,
                                       msg=lambda m: f"step {step} {k}: {m}")
        torch.testing.assert_close(loss_g.cpu().reshape(-1), loss_o.reshape(-1), rtol=1e-4, atol=1e-5)
    # parameters after 6 optimiser steps and the EMA teacher
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        assert n1 == n2
        if n1.endswith("lin_l.bias"):
            # a bias in front of BatchNorm has an exactly-zero true gradient: what reaches AdamW is
            # rounding noise, which Adam's normalisation turns into +-lr steps on both sides
            continue
        torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-4, atol=2e-5, msg=lambda m: f"{n1}: {m}")


def test_moe_encoder_vs_oracle(dev):
    """--moe --moe_layers all (reference encoder.py:109-129,292-309): reversed-direction plain
    mean aggregation + K-expert einsum + softmax routing; eval mode (no Gumbel noise) and the
    mixture layer alone in train mode with gradients."""
    from stem_gnn_amd.model.encoder import Encoder, MixtureSageLayer
    N, E, D = 300, 2500, 32
    torch.manual_seed(2)
    oe = O.OracleEncoder(D, D, 2, normalize="batch", dropout=0.0, moe=True, num_experts=3, moe_layers="all")
    ge = Encoder(D, D, nn.ReLU, 2, normalize="batch", dropout=0.0, moe=True, num_experts=3, moe_layers="all")
    ge.load_state_dict(oe.state_dict())
    ge = ge.to(dev)
    x = torch.randn(N, D)
    ei = torch.randint(0, N, (2, E))
    oe.eval(); ge.eval()
    with torch.no_grad():
        torch.testing.assert_close(ge(x.to(dev), ei.to(dev)).cpu(), oe(x, ei), rtol=1e-4, atol=1e-4)
    ol, gl = oe.layers[0], ge.layers[0]
    xr = x.clone().requires_grad_(True)
    xg = x.to(dev).requires_grad_(True)
    w = torch.randn(N, 3, D)
    (ol(xr, ei) * w).sum().backward()
    (gl(xg, ei.to(dev)) * w.to(dev)).sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gl.weights.grad.cpu(), ol.weights.grad, rtol=1e-4, atol=1e-3)


def test_finetune_consumer_flow(dev):
    """The downstream contract of reference finetune.py:131-181 / utils/others.py:160-171 /
    model/ft_model.py:90-103: VectorQuantize(kmeans_init=True) -> dummy forward on randn(100, dim)
    flips `initted` -> load_state_dict of a pretrained vq -> frozen, eval-free use inside
    TaskModel.get_lin_logits (decoder over the per-head codes)."""
    from stem_gnn_amd.model.vq import VectorQuantize
    D, H, K = 64, 4, 32
    torch.manual_seed(3)
    ovq = O.OracleVectorQuantize(D, K, D, H, commitment_weight=10.0, orthogonal_reg_weight=1.0,
                                 orthogonal_reg_max_codes=32, ema_update=False)
    pretrained = ovq.state_dict()
    vq = VectorQuantize(dim=D, codebook_size=K, codebook_dim=D, heads=H, separate_codebook_per_head=True, decay=0.8,
                        commitment_weight=10, use_cosine_sim=True, orthogonal_reg_weight=1, orthogonal_reg_max_codes=32,
                        kmeans_init=True, ema_update=False).to(dev)
    assert float(vq._codebook.initted) == 0.0
    vq(torch.randn(100, D, device=dev))            # others.py:168-169: k-means init on a dummy batch
    assert float(vq._codebook.initted) == 1.0 and bool(torch.isfinite(vq._codebook.embed).all())
    vq.load_state_dict(pretrained)                  # others.py:170
    for p in vq.parameters():                       # finetune.py:179-181 freeze_params
        p.requires_grad_(False)
    z = torch.randn(300, D)
    zg = z.to(dev).requires_grad_(True)
    vq.train()                                      # TaskModel trains with the frozen vq in train mode
    q, ind, loss, codes = vq(zg)
    ovq.train()
    qr, ir, lr, cr = ovq(z, ortho_ids=torch.arange(K))
    with torch.no_grad():
        x = torch.nn.functional.normalize(ovq.project_in(z).view(300, H, D), dim=-1)
        top2 = torch.einsum("nhd,hcd->nhc", x, ovq._codebook.embed).topk(2, dim=-1).values
    flips = assert_indices_match(ind.cpu(), ir, top2[..., 0] - top2[..., 1])
    assert tuple(codes.shape) == (300, H * D)       # ft_model.py:94: decoder(codes).reshape(-1, H, C)
    if flips == 0:
        torch.testing.assert_close(codes.detach().cpu(), cr.detach(), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(q.detach().cpu(), qr.detach(), rtol=1e-4, atol=1e-5)
    head = nn.Linear(H * D, H * 7).to(dev)
    logits = head(codes).reshape(-1, H, 7).mean(1)
    torch.nn.functional.cross_entropy(logits, torch.randint(0, 7, (300,), device=dev)).backward()
    assert zg.grad is not None and bool(torch.isfinite(zg.grad).all())
    assert tuple(vq.codebook.shape) == (H, K, D)


def test_prefetch_loader_yields_the_same_batches(dev):
    """PrefetchLoader samples one batch ahead on a side stream: same batches, same order, usable on the current
    stream right away (values checked after more allocator traffic on both streams)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader, PrefetchLoader
    from stem_gnn_amd.data.synthetic import make_graph
    g = make_graph(20_000, 200_000, 32, 3, kind="U", device=dev)
    node_labels = torch.arange(20_000, device=dev) % 7   # a node-level attribute: batches carry y = labels[n_id]

    def make_loader():
        s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [5, 5], seed=3)
        return NeighborLoader(s, torch.arange(4096, device=dev), 512, shuffle=True, seed=1, y=node_labels)

    plain = [(b.n_id.clone(), b.edge_index.clone(), b.xe.clone()) for b in make_loader()]

    def prepare(b):
        b.feat = ops.gather_rows(g.node_text_feat, b.x.contiguous())
        b.graph.ensure_transpose()

    got = []
    for b in PrefetchLoader(make_loader(), dev, prepare):
        junk = torch.randn(1 << 20, device=dev).sum()          # allocator + stream traffic between hand-over and use
        assert torch.equal(b.y, node_labels[b.n_id])
        got.append((b.n_id, b.edge_index, b.xe, b.feat, b.graph.rowptr_t, junk))
    torch.cuda.synchronize()
    assert len(got) == len(plain) == 8
    for (n_id, ei, xe, feat, rpt, _), (n0, e0, x0) in zip(got, plain):
        assert torch.equal(n_id, n0) and torch.equal(ei, e0) and torch.equal(xe, x0)
        assert torch.equal(feat, g.node_text_feat[n_id])
        assert int(rpt[-1]) == ei.size(1)
    # a loader without iter_pending() (any iterable of batches) takes the one-deep path
    again = [b.n_id for b in PrefetchLoader(list(make_loader()), dev)]
    assert len(again) == 8 and all(torch.equal(a, n0) for a, (n0, _, _) in zip(again, plain))


def test_pretrain_step_on_sampler_batch_matches_oracle(dev):
    """The step on a HIP-sampler batch (GraphStructure with active_rows: only the expanded, leading nodes receive
    edges -> the layer products skip the zero part of the aggregate and its gradient) against the CPU oracle on the
    same batch with the draws replayed, two optimiser steps."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import pretrain_step, default_params
    D, L, H, K = 64, 2, 4, 64
    g = make_graph(5000, 60000, D, 4, kind="U", device=dev)
    s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [6, 6], seed=2)
    b = s.sample(torch.randperm(5000, device=dev)[:96])
    gs = b.graph
    n, e = b.n_id.numel(), b.edge_index.size(1)
    assert gs.active_rows is not None and 96 <= gs.active_rows < n
    indeg = torch.bincount(b.edge_index[1], minlength=n)
    assert int(indeg[gs.active_rows:].sum()) == 0 and int(indeg[:gs.active_rows].sum()) == e
    om, gm = make_models(D, L, H, K, D, dev)
    params = default_params()
    x = g.node_text_feat[b.n_id]
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-4, weight_decay=1e-5)
    ops.manual_seed(21)
    x_cpu, ei_cpu, ea_cpu = x.cpu(), b.edge_index.cpu(), g.edge_text_feat[b.xe].cpu()
    for step in range(2):
        loss_g, losses_g, draws = pretrain_step(gm, opt_g, None, params, x, gs, EdgeTypeAttr(g.edge_text_feat, b.xe), 96)
        aug = gm  # noqa: F841
        cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
        loss_o, losses_o, _ = O.pretrain_step(om, opt_o, None, params, x_cpu, ei_cpu, ea_cpu, 96, cpu_draws)
        for k in losses_o:
            torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=1e-4, atol=1e-5,
                                       msg=lambda m: f"step {step} {k}: {m}")
        torch.testing.assert_close(loss_g.cpu().reshape(-1), loss_o.reshape(-1), rtol=1e-4, atol=1e-5)
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        if "lin_l.bias" in n1 or n1.startswith("sem_encoder"):
            continue
        torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-3, atol=3e-4, msg=lambda m: f"{n1}: {m}")


def test_moe_encoder_train_mode_matches_oracle(dev):
    """--moe in TRAINING mode (reference encoder.py:292-309, 202-204): Gumbel-softmax routing and the environment
    regulariser.  The HIP run's Gumbel draws are recorded (Encoder.last_gumbel_noise) and replayed through the oracle;
    then the oracle's path is checked the other way round with injected noise.  env_reg_loss is non-zero and matches,
    and so do the output and every gradient."""
    from stem_gnn_amd.model.encoder import Encoder
    N, E, D = 300, 2500, 32
    torch.manual_seed(4)
    oe = O.OracleEncoder(D, D, 2, normalize="batch", dropout=0.0, moe=True, num_experts=3, moe_layers="all", tau=0.7)
    ge = Encoder(D, D, nn.ReLU, 2, normalize="batch", dropout=0.0, moe=True, num_experts=3, tau=0.7, moe_layers="all")
    ge.load_state_dict(oe.state_dict())
    ge = ge.to(dev)
    x = torch.randn(N, D)
    ei = torch.randint(0, N, (2, E))
    w = torch.randn(N, D)
    oe.train(); ge.train()
    for inject in (False, True):
        noise = [-torch.empty(N, 3).exponential_().log() for _ in range(2)] if inject else None
        ge.gumbel_noise = None if noise is None else [t.to(dev) for t in noise]
        ge.zero_grad(); oe.zero_grad()
        zg = ge(x.to(dev), ei.to(dev))
        reg_g = ge.get_env_reg()
        used = [t.cpu() for t in ge.last_gumbel_noise]
        assert len(used) == 2 and all(tuple(t.shape) == (N, 3) for t in used)
        if inject:
            assert all(torch.equal(a, b) for a, b in zip(used, noise))
        zo = oe(x, ei, gumbel_noise=used)
        reg_o = oe.get_env_reg()
        assert abs(float(reg_o)) > 1e-3  # the regulariser is live in this mode
        torch.testing.assert_close(reg_g.cpu().reshape(-1), reg_o.reshape(-1), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(zg.detach().cpu(), zo.detach(), rtol=1e-4, atol=1e-4)
        ((zg * w.to(dev)).sum() + 3.0 * reg_g.sum()).backward()
        ((zo * w).sum() + 3.0 * reg_o.sum()).backward()
        for (n1, p1), (n2, p2) in zip(oe.named_parameters(), ge.named_parameters()):
            assert n1 == n2
            scale = max(p1.grad.abs().max().item(), 1.0)
            torch.testing.assert_close(p2.grad.cpu(), p1.grad, rtol=1e-3, atol=1e-4 * scale, msg=lambda m: f"{n1}: {m}")
    usage = ge.get_moe_usage()
    assert [u["layer"] for u in usage] == [0, 1] and all(abs(sum(u["avg_prob"]) - 1.0) < 1e-4 for u in usage)
    assert all(abs(sum(u["top1_frac"]) - 1.0) < 1e-6 for u in usage) and ge.get_moe_usage() == []


def _ddp_worker(rank, world, port, out):
    """One rank of the 2-rank HIP data-parallel test (both ranks on cuda:0, gloo for the collective)."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from stem_gnn_amd import ops, parallel
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    D, L, H, K = 64, 2, 4, 64
    _, gm = make_models(D, L, H, K, D, dev)
    for p in gm.sem_encoder.parameters():
        p.requires_grad_(False)
    params = default_params()
    fwd = parallel.wrap_ddp(gm, 0)
    gen = torch.Generator().manual_seed(100)
    batches = []
    for r in range(world):
        N, E = 500 + 40 * r, 4000 + 100 * r
        x = torch.nn.functional.normalize(torch.randn(N, D, generator=gen), dim=-1)
        half = torch.randint(0, N, (2, E // 2), generator=gen)
        ei = torch.cat([half, half.flip(0)], dim=1)
        table = torch.nn.functional.normalize(torch.randn(4, D, generator=gen), dim=-1)
        et = torch.randint(0, 4, (E,), generator=gen)
        batches.append((x, ei, table, et))
    x, ei, table, et = batches[rank]

    class NoStep:  # gradients only: the optimiser must not move the weights
        def zero_grad(self, set_to_none=True):
            for p in gm.parameters():
                p.grad = None

        def step(self, *a, **k):
            pass

    res = {}

    def capture():  # runs between backward and clipping (pretrain_step's grad_sync hook): the reducer has finished
        torch.cuda.synchronize()
        res.update({n: p.grad.detach().cpu().clone() for n, p in gm.named_parameters() if p.grad is not None})

    ops.manual_seed(1000 + rank)
    gm.train()
    pretrain_step(gm, NoStep(), None, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), 128,
                  record_draws=False, forward_fn=fwd, grad_sync=capture)
    torch.cuda.synchronize()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_ddp_averages_gradients(dev):
    """Two ranks of the HIP PretrainModel under wrap_ddp (DistributedDataParallel; both ranks on the one card, gloo
    moving the buckets): every rank ends with the same gradients, and they are the mean of the two ranks' own
    gradients as one process computes them for the same two batches and draws (custom autograd phases under the
    reducer, teacher excluded, BatchNorm statistics per rank)."""
    import socket
    import torch.multiprocessing as mp
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    g0, g1 = out[0], out[1]
    assert set(g0) == set(g1) and not any(k.startswith("sem_encoder") for k in g0)
    for k in g0:
        torch.testing.assert_close(g0[k], g1[k], rtol=0, atol=0, msg=lambda m: f"{k}: ranks disagree: {m}")
    # the same two batches in one process, gradients accumulated by hand
    torch.manual_seed(0)
    D, L, H, K = 64, 2, 4, 64
    _, gm = make_models(D, L, H, K, D, dev)
    params = default_params()
    gen = torch.Generator().manual_seed(100)
    acc = {}
    for r in range(2):
        N, E = 500 + 40 * r, 4000 + 100 * r
        x = torch.nn.functional.normalize(torch.randn(N, D, generator=gen), dim=-1)
        half = torch.randint(0, N, (2, E // 2), generator=gen)
        ei = torch.cat([half, half.flip(0)], dim=1)
        table = torch.nn.functional.normalize(torch.randn(4, D, generator=gen), dim=-1)
        et = torch.randint(0, 4, (E,), generator=gen)

        class NoStep:
            def zero_grad(self, set_to_none=True):
                for p in gm.parameters():
                    p.grad = None

            def step(self, *a, **k):
                pass

        def capture():
            for n, p in gm.named_parameters():
                if p.grad is not None and not n.startswith("sem_encoder"):
                    acc[n] = acc.get(n, 0) + p.grad.detach().cpu() / 2

        ops.manual_seed(1000 + r)
        gm.train()
        # BatchNorm buffers and the EMA teacher move during a step; every rank of the DDP run starts from the initial state
        state = {k: v.clone() for k, v in gm.state_dict().items()}
        pretrain_step(gm, NoStep(), None, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), 128,
                      record_draws=False, grad_sync=capture)
        gm.load_state_dict(state)
    for k in g0:
        scale = max(acc[k].abs().max().item(), 1e-6)
        torch.testing.assert_close(g0[k], acc[k], rtol=1e-4, atol=1e-5 * max(scale, 1.0), msg=lambda m: f"{k}: {m}")


def _ddp_batches(world, steps, D):
    """The (rank, step) batches of the two-step data-parallel test, from one shared generator."""
    gen = torch.Generator().manual_seed(321)
    out = {}
    for r in range(world):
        for s_ in range(steps):
            N, E = 480 + 30 * r + 10 * s_, 3600 + 200 * r + 100 * s_
            x = torch.nn.functional.normalize(torch.randn(N, D, generator=gen), dim=-1)
            half = torch.randint(0, N, (2, E // 2), generator=gen)
            ei = torch.cat([half, half.flip(0)], dim=1)
            table = torch.nn.functional.normalize(torch.randn(4, D, generator=gen), dim=-1)
            et = torch.randint(0, 4, (E,), generator=gen)
            out[(r, s_)] = (x, ei, table, et)
    return out


def _ddp_step_worker(rank, world, port, out, mode="ddp"):
    """One rank of the two-step test: wrap_ddp + FusedAdamW (clip factor folded into its gradient read) + the cosine
    schedule + the teacher EMA -- the optimiser reads the reducer's bucket views (gradient_as_bucket_view)."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from stem_gnn_amd import ops, parallel
    from stem_gnn_amd._lib import lib
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    from stem_gnn_amd.utils.others import get_scheduler
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    lib.stemgnn_set_deterministic(1)  # fixed-order decoder scatters: the gradients are a pure function of the inputs
    D, L, H, K = 64, 2, 4, 64
    _, gm = make_models(D, L, H, K, D, dev)
    params = default_params()
    fwd, sync = None, None
    if mode == "ddp":
        fwd = parallel.wrap_ddp(gm, 0)
    else:  # bench.py's default exchange for --gpus N: one fused copy + one all-reduce, p.grad = views of the flat buffer
        for p in gm.sem_encoder.parameters():
            p.requires_grad_(False)
        sync = parallel.FlatGradSync(gm.parameters())
    opt = ops.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=1e-2)
    sched = get_scheduler(opt, True, 50)
    batches = _ddp_batches(world, 2, D)
    ops.manual_seed(1000 + rank)
    gm.train()
    for s_ in range(2):
        x, ei, table, et = batches[(rank, s_)]
        pretrain_step(gm, opt, sched, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), 128,
                      record_draws=False, forward_fn=fwd, grad_sync=sync)
    torch.cuda.synchronize()
    assert isinstance(opt, ops.FusedAdamW) and all(opt.state[p]["step"] == 2 for p in gm.parameters() if p.requires_grad)
    out[rank] = {n: p.detach().cpu().clone() for n, p in gm.named_parameters()}
    # the reducer's own record: after the first step it re-cut the gradient into the buckets wrap_ddp's cap asks for
    # (its statistics are refreshed every few iterations: one more step, after the parameters have been handed back)
    if mode == "ddp":
        x, ei, table, et = batches[(rank, 0)]
        pretrain_step(gm, opt, sched, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), 128,
                      record_draws=False, forward_fn=fwd)
        torch.cuda.synchronize()
        log = fwd._get_ddp_logging_data()
        out[("buckets", rank)] = [int(v) for v in str(log.get("rebuilt_bucket_sizes", "")).split(",") if v.strip()]
    else:
        out[("buckets", rank)] = [int(sync.flat.numel() * 4)]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_two_rank_hip_ddp_two_optimizer_steps(dev, mode):
    """the round-2 reducer test used a no-op optimiser.  Here two ranks take TWO real steps
    (FusedAdamW reading gradient_as_bucket_view buckets, clip factor folded in, per-batch cosine schedule, teacher
    EMA).  (i) Both ranks end with bit-equal parameters, teacher included.  (ii) They match one process that computes
    both ranks' gradients for the same batches and draws, averages them by hand and applies the same update: 1e-6
    absolute (deterministic decoder scatters on both sides, so the gradients are the same numbers and Adam's
    sign-like first steps cannot turn rounding noise into +-lr differences); with lr = 1e-3 two steps move every
    trained parameter by up to 2e-3, three orders above the tolerance."""
    import socket
    import torch.multiprocessing as mp
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step, _trainable
    from stem_gnn_amd.utils.others import get_scheduler
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    procs = [ctx.Process(target=_ddp_step_worker, args=(r, 2, port, out, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    p0, p1 = out[0], out[1]
    assert set(p0) == set(p1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), f"{k}: the ranks' parameters differ after two steps"
    # round-3 review, item 6: several readiness points per step -- the second step reduced at least three buckets (the
    # heads', the quantiser's and the encoder's stretch of the backward), so reductions overlap the backward behind them
    # ("flat": bench.py's default exchange -- one flat all-reduce of the whole trainable gradient)
    assert out[("buckets", 0)] == out[("buckets", 1)] and len(out[("buckets", 0)]) >= (3 if mode == "ddp" else 1), out[("buckets", 0)]

    # ---- the same two global steps in one process
    D, L, H, K = 64, 2, 4, 64
    _, gm = make_models(D, L, H, K, D, dev)
    init = {n: p.detach().cpu().clone() for n, p in gm.named_parameters()}
    for p in gm.sem_encoder.parameters():
        p.requires_grad_(False)
    params = default_params()
    opt = ops.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=1e-2)
    sched = get_scheduler(opt, True, 50)
    batches = _ddp_batches(2, 2, D)
    keys = {r: [(1000 + r) & 0xFFFFFFFFFFFFFFFF, 0] for r in range(2)}  # each rank's Philox (seed, call counter)

    class NoStep:
        def zero_grad(self, set_to_none=True):
            for p in gm.parameters():
                p.grad = None

        def step(self, *a, **k):
            pass

    prev = lib.stemgnn_set_deterministic(1)
    try:
        gm.train()
        for s_ in range(2):
            acc = {}

            def capture():
                for n, p in gm.named_parameters():
                    if p.grad is not None:
                        acc[n] = acc.get(n, 0) + p.grad.detach().clone() / 2

            for r in range(2):
                x, ei, table, et = batches[(r, s_)]
                state = {k: v.clone() for k, v in gm.state_dict().items()}
                ops._keys.seed, ops._keys.counter = keys[r]
                pretrain_step(gm, NoStep(), None, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)),
                              128, record_draws=False, grad_sync=capture)
                keys[r] = [ops._keys.seed, ops._keys.counter]
                gm.load_state_dict(state)  # the gradient pass moved BatchNorm buffers and the teacher: undo
            for n, p in gm.named_parameters():
                p.grad = acc.get(n)
            grads = [p.grad for p in _trainable(gm) if p.grad is not None]
            opt.step(grad_coef=ops.grad_norm_coef(grads, 1.0)[1:])  # pretrain_step's clip + AdamW
            sched.step()
            gm.ema_update_sem_encoder(decay=params["sem_encoder_decay"])
        torch.cuda.synchronize()
    finally:
        lib.stemgnn_set_deterministic(prev)
    moved = 0.0
    for n, p in gm.named_parameters():
        torch.testing.assert_close(p0[n], p.detach().cpu(), rtol=0, atol=1e-6, msg=lambda m: f"{n}: {m}")
        moved = max(moved, float((p0[n] - init[n]).abs().max()))
    assert moved > 1e-3  # the steps were real


def test_pretrain_step_on_a_multi_dataset_mix_batch(dev):
    """BASELINE config 5 plumbing (`--pretrain_dataset all`): a union of nine member graphs (data/multi.py), the
    per-epoch weighted seed list, HIP-sampled batches whose edges never leave a member, 264 edge types (the type table
    does not fit LDS: K1's global-table mode) -- and the step on such a batch against the CPU oracle."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.multi import mix_weights, synthetic_mix
    from stem_gnn_amd.data.sampler import HipNeighborSampler, MixLoader
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    D, L, H, K = 64, 2, 4, 64
    u = synthetic_mix("all", dim=D, device=dev, scale=0.01, seed=5)
    assert len(u.names) == 9 and u.edge_text_feat.size(0) == 1 + 1 + 1 + 1 + 11 + 237 + 4 + 4 + 4
    weights = list(mix_weights("all").values())
    s = HipNeighborSampler(u.edge_index, u.xe, u.num_nodes, u.x, u.node_text_feat, u.edge_text_feat, [6, 6], seed=2)
    loader = MixLoader(s, u.ptr, weights, 128, seed=3, device=dev)
    it = iter(loader)
    b = next(it)
    sizes = (u.ptr[1:] - u.ptr[:-1]).tolist()
    assert loader.num_seeds == sum(int(w) * n + int((w - int(w)) * n) for w, n in zip(weights, sizes))
    member = torch.bucketize(b.n_id.cpu(), u.ptr[1:], right=True)
    ei = b.edge_index.cpu()
    assert torch.equal(member[ei[0]], member[ei[1]])  # sampled edges stay inside their member graph
    assert len(set(member[:128].tolist())) >= 3        # a batch mixes members
    om, gm = make_models(D, L, H, K, D, dev)
    params = default_params()
    x = ops.gather_rows(u.node_text_feat, b.x.contiguous())
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-4, weight_decay=1e-5)
    ops.manual_seed(8)
    loss_g, losses_g, draws = pretrain_step(gm, opt_g, None, params, x, b.graph, EdgeTypeAttr(u.edge_text_feat, b.xe), 128)
    cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
    loss_o, losses_o, _ = O.pretrain_step(om, opt_o, None, params, x.cpu(), ei, u.edge_text_feat[b.xe].cpu(), 128, cpu_draws)
    for k in losses_o:
        torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=1e-4, atol=1e-5,
                                   msg=lambda m: f"{k}: {m}")
    # a second epoch redraws the fractional members' share
    first = loader._epoch_nodes()
    loader.epoch += 1
    second = loader._epoch_nodes()
    assert first.numel() == second.numel() and not torch.equal(first, second)


def test_pretrain_step_with_bf16_feature_storage(dev):
    """BASELINE config 5's bf16 option as this library builds it: node features and the layer outputs that the next
    layer reads are STORED as bf16 (half the bytes for K1's gather, the layer products' activation operand and K2),
    all arithmetic, the aggregates, z, the VQ core and every gradient stay fp32.  Two bars, both written here:
      * 1e-4 against the oracle with the SAME storage roundings emulated (round-to-nearest-even at the same points);
      * 2e-2 against the plain fp32 oracle: what the storage mode itself costs (bf16 has 8 significant bits)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    N, E, D, L, H, K, bs = 600, 5000, 64, 2, 4, 64, 200
    om, gm = make_models(D, L, H, K, D, dev)
    om32, _ = make_models(D, L, H, K, D, dev)
    om.encoder.bf16_storage = om.sem_encoder.bf16_storage = True
    params = default_params()
    torch.manual_seed(12)
    x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1).bfloat16()   # the stored features
    half = torch.randint(0, N, (2, E // 2))
    ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
    table = torch.nn.functional.normalize(torch.randn(4, D), dim=-1)
    et = torch.randint(0, 4, (E,))
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_32 = torch.optim.AdamW(om32.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_g = torch.optim.AdamW(gm.parameters(), lr=1e-4, weight_decay=1e-5)
    ops.manual_seed(5)
    xg = x.to(dev)
    assert xg.dtype == torch.bfloat16
    for step in range(2):
        loss_g, losses_g, draws = pretrain_step(gm, opt_g, None, params, xg, ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), bs)
        cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
        loss_o, losses_o, _ = O.pretrain_step(om, opt_o, None, params, x.float(), ei, table[et], bs, cpu_draws)
        # the plain-fp32 oracle sees different activations: its arg-max is its own (no near-tie replay)
        loss_f, losses_f, _ = O.pretrain_step(om32, opt_32, None, params, x.float(), ei, table[et], bs,
                                              {k: v for k, v in cpu_draws.items() if k != "vq_indices"})
        for k in losses_o:
            torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=1e-4, atol=1e-5,
                                       msg=lambda m: f"step {step} {k} (storage emulated): {m}")
            torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_f[k].reshape(-1), rtol=2e-2, atol=1e-3,
                                       msg=lambda m: f"step {step} {k} (plain fp32 oracle): {m}")
    # the modules really ran on bf16 storage: the student's first-layer output is a bf16 tensor's worth of values
    with torch.no_grad():
        z = gm.encoder(xg, ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)))
    assert z.dtype == torch.float32 and tuple(z.shape) == (N, D)
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        if "lin_l.bias" in n1 or n1.startswith("sem_encoder"):
            continue
        torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-3, atol=3e-4, msg=lambda m: f"{n1}: {m}")


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_pretrain_step_with_bf16_gemms(dev, storage):
    """BASELINE config 5 as written ("bf16"; SURVEY.md section 7 step 7: autocast for the K3 / K5 products, the VQ core
    fp32): ``ops.linear_set_mode(2)`` runs every Linear of the path -- lin_l / lin_r, project_in / project_out, the
    decoders, the semantic projector, forward and both backward products -- as ONE bf16 matrix pass on operands rounded
    to bf16, fp32 accumulation; the quantiser's similarity / arg-max stays exact (vq.py:623,634).  Two bars:
      * 2e-4 (+1e-5) against the oracle that rounds the same operands at the same points (O.bf16_gemms): the two sides
        differ by fp32 summation order, and where that moves a value across a bf16 rounding boundary, by one bf16 ulp
        of that element; near-ties of the arg-max are replayed up to 1e-3 for the same reason and counted;
      * 3e-2 against the plain fp32 oracle: what the mode itself costs (8 significant bits per operand)."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import default_params, pretrain_step
    N, E, D, L, H, K, bs = 600, 5000, 64, 2, 4, 64, 200
    om, gm = make_models(D, L, H, K, D, dev)
    om32, _ = make_models(D, L, H, K, D, dev)
    bf = storage == "bf16"
    if bf:
        om.encoder.bf16_storage = om.sem_encoder.bf16_storage = True
    params = default_params()
    torch.manual_seed(13)
    x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1)
    if bf:
        x = x.bfloat16()
    half = torch.randint(0, N, (2, E // 2))
    ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
    table = torch.nn.functional.normalize(torch.randn(4, D), dim=-1)
    et = torch.randint(0, 4, (E,))
    opt_o = torch.optim.AdamW(om.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_32 = torch.optim.AdamW(om32.parameters(), lr=1e-4, weight_decay=1e-5)
    opt_g = ops.FusedAdamW(gm.parameters(), lr=1e-4, weight_decay=1e-5)
    ops.manual_seed(6)
    prev = ops.linear_set_mode(2)
    ties = 0
    try:
        assert ops.linear_set_mode(-1) == 2
        for step in range(3):
            loss_g, losses_g, draws = pretrain_step(gm, opt_g, None, params, x.to(dev), ei.to(dev),
                                                    EdgeTypeAttr(table.to(dev), et.to(dev)), bs)
            cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
            cpu_draws["vq_tie_tol"] = 1e-3
            with O.bf16_gemms():
                loss_o, losses_o, _ = O.pretrain_step(om, opt_o, None, params, x.float(), ei, table[et], bs, cpu_draws)
            ties += om.vq.last_tie_adopted
            loss_f, losses_f, _ = O.pretrain_step(om32, opt_32, None, params, x.float(), ei, table[et], bs,
                                                  {k: v for k, v in cpu_draws.items() if k not in ("vq_indices", "vq_tie_tol")})
            for k in losses_o:
                torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_o[k].reshape(-1), rtol=2e-4, atol=1e-5,
                                           msg=lambda m: f"step {step} {k} (bf16 GEMMs emulated): {m}")
                torch.testing.assert_close(losses_g[k].cpu().reshape(-1), losses_f[k].reshape(-1), rtol=3e-2, atol=2e-3,
                                           msg=lambda m: f"step {step} {k} (plain fp32 oracle): {m}")
    finally:
        ops.linear_set_mode(prev)
    assert ties <= 12, ties
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        if "lin_l.bias" in n1 or n1.startswith("sem_encoder"):
            continue
        torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-3, atol=3e-4, msg=lambda m: f"{n1}: {m}")


@pytest.mark.parametrize("m,k1,k2,n,rows", [(1000, 128, 0, 128, -1), (777, 64, 32, 96, -1), (2000, 128, 128, 128, 300),
                                            (4097, 768, 0, 128, -1), (1300, 96, 0, 3072, -1)])
def test_bf16_gemm_mode_is_the_product_of_the_rounded_operands(dev, m, k1, k2, n, rows):
    """Kernel-level statement of mode 2: y = round(x1) round(w1)^T (+ round(x2) round(w2)^T) + b, dx = round(dy) round(w),
    dw = round(dy)^T round(x), db = colsum(dy), all accumulated in fp32 -- against torch on the rounded operands in fp64
    (1e-5 of the largest output: fp32 accumulation only)."""
    from stem_gnn_amd import ops
    torch.manual_seed(m + n)
    r = lambda t: t.bfloat16().double()  # noqa: E731
    a = torch.randn(m, k1, device=dev) * (1 + 3 * torch.rand(m, 1, device=dev))
    w = torch.randn(n, k1, device=dev) * 0.2
    a2 = torch.randn(m, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.2 if k2 else None
    b = torch.randn(n, device=dev)
    if rows >= 0:
        a[rows:] = 0
    dy = torch.randn(m, n, device=dev)
    prev = ops.linear_set_mode(2)
    try:
        y, _, _ = ops.linear_fwd(a, w, a2, w2, b, False, rows)
        dx = ops.linear_bwd_data(dy, w)
        dw, db = ops.linear_bwd_weight(dy, a, True)
    finally:
        ops.linear_set_mode(prev)
    ref = r(a) @ r(w).t() + (r(a2) @ r(w2).t() if k2 else 0) + b.double()
    tol = lambda t: 1e-5 * float(t.abs().max())  # noqa: E731
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=tol(ref))
    ref_dx = r(dy) @ r(w)
    torch.testing.assert_close(dx.double(), ref_dx, rtol=1e-5, atol=tol(ref_dx))
    ref_dw = r(dy).t() @ r(a)
    torch.testing.assert_close(dw.double(), ref_dw, rtol=1e-5, atol=tol(ref_dw))
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-5, atol=1e-4 * float(dy.abs().sum(0).max()))


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "bf16"])
@pytest.mark.parametrize("m,k1,k2,n,rows", [(32768, 768, 0, 768, -1), (20000, 512, 512, 1024, 4000), (9000, 512, 0, 3072, -1),
                                            (200_001, 256, 0, 256, -1)])
def test_large_products_on_the_bigtile_core(dev, mode, m, k1, k2, n, rows):
    """The large products (>= 8 192 rows, feature extents >= 256: the D = 768 configurations) run on the big-tile core
    (csrc/bigtile.hip: cut pass + one 256 x 256 x 64 bf16 MFMA GEMM) in the exact mode (six exact piece products:
    fp32-accurate, checked against fp64 of the fp32 operands) and in the bf16 GEMM mode (rounded operands, checked against
    fp64 of the rounded operands) -- AND against the 128-row tile kernels (core switched off): y with bias, the
    two-operand form with zero leading rows, the BatchNorm column sums, backward-data, weight and bias gradients.  Row
    counts that are no multiple of the 256-row tile, a 12-tile-wide output, a one-tile output."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    torch.manual_seed(m + n)
    r = (lambda t: t.bfloat16().double()) if mode == 2 else (lambda t: t.double())  # noqa: E731
    a = torch.randn(m, k1, device=dev) * (1 + 3 * torch.rand(m, 1, device=dev))
    w = torch.randn(n, k1, device=dev) * 0.2
    a2 = torch.randn(m, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.2 if k2 else None
    b = torch.randn(n, device=dev)
    if rows >= 0:
        a[rows:] = 0
    dy = torch.randn(m, n, device=dev)
    prev = ops.linear_set_mode(mode)
    out = {}
    try:
        for core_on in (1, 0):
            was = ops.linear_set_bigtile(core_on)
            served, missed = lib.stemgnn_linear_bigtile_calls(), lib.stemgnn_linear_bigtile_fallbacks()
            y, part, blocks = ops.linear_fwd(a, w, a2, w2, b, True, rows)
            out[core_on] = (y, part[:blocks].double().sum(0), ops.linear_bwd_data(dy, w), *ops.linear_bwd_weight(dy, a, True))
            took = lib.stemgnn_linear_bigtile_calls() - served  # forward, backward-data, weight gradient
            assert took == (3 if core_on else 0), took
            assert lib.stemgnn_linear_bigtile_fallbacks() == missed  # the arena was there: nothing fell back
            ops.linear_set_bigtile(was)
    finally:
        ops.linear_set_mode(prev)
    ref = r(a) @ r(w).t() + (r(a2) @ r(w2).t() if k2 else 0) + b.double()
    refs = (ref, torch.stack([ref.sum(0), (ref * ref).sum(0)]), r(dy) @ r(w), r(dy).t() @ r(a), dy.double().sum(0))
    for name, got_core, got_tile, want in zip(("y", "column sums", "dx", "dw", "db"), out[1], out[0], refs):
        tol = 2e-5 * float(want.abs().max())
        torch.testing.assert_close(got_core.double(), want, rtol=1e-5, atol=tol, msg=lambda s_: f"{name} (core): {s_}")
        torch.testing.assert_close(got_core.double(), got_tile.double(), rtol=1e-5, atol=tol,
                                   msg=lambda s_: f"{name} (core vs tile kernel): {s_}")
    # an operand stored as bf16 (feature_kind 1: C5's hidden activations) is its own h piece
    from stem_gnn_amd._lib import check
    xb = a.bfloat16().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    prev = ops.linear_set_mode(mode)
    try:
        dws = []
        for core_on in (1, 0):
            was = ops.linear_set_bigtile(core_on)
            dw = torch.empty(n, k1, device=dev)
            ws = torch.empty(int(lib.stemgnn_linear_bwd_weight_workspace_bytes(m, n, k1)), dtype=torch.uint8, device=dev)
            check(lib.stemgnn_linear_bwd_weight_k(dy.data_ptr(), xb.data_ptr(), 1, m, n, k1, dw.data_ptr(), None, ws.data_ptr(),
                                                  ws.numel(), st))
            dws.append(dw)
            ops.linear_set_bigtile(was)
    finally:
        ops.linear_set_mode(prev)
    want = r(dy).t() @ xb.double()
    torch.testing.assert_close(dws[0].double(), want, rtol=1e-5, atol=2e-5 * float(want.abs().max()))
    torch.testing.assert_close(dws[0], dws[1], rtol=1e-5, atol=2e-5 * float(want.abs().max()))


@pytest.mark.parametrize("case", ["normal", "row_scales", "in_row_range", "outlier", "tiny_and_huge_rows", "zero_rows"])
def test_pair_format_products_are_fp32_accurate(dev, case):
    """The exact mode's pair format (csrc/bigtile.hip: two fp16 pieces of rows scaled by a power of two, three matrix
    passes) against fp64, for the forward and the backward-data product, on operands chosen to strain it: rows of very
    different magnitudes, eight decades of range INSIDE a row, one outlier of 1e4 among 1e-4 values, rows at 1e-30 and
    1e+30, all-zero rows.  Yardstick: the error of the fp32-MFMA kernels (mode 0: an fp32 fma chain -- what "fp32" means
    for this product) and of the three-bf16-piece form on the same operands, every error taken relative to
    sum_k |x_k| |w_k| of its element (the norm-wise measure an fp32 sum is judged by).  The pair format must be no worse
    than twice the fp32 chain, level with the bf16-piece form, and finite wherever fp64's results are."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    torch.manual_seed(11)
    m, k, n = 32768, 768, 512
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    if case == "row_scales":
        x *= 10.0 ** (torch.rand(m, 1, device=dev) * 12 - 6)
        w *= 10.0 ** (torch.rand(n, 1, device=dev) * 6 - 3)
    elif case == "in_row_range":
        x *= 10.0 ** (-8 * torch.rand(m, k, device=dev))
        w *= 10.0 ** (-4 * torch.rand(n, k, device=dev))
    elif case == "outlier":
        x *= 1e-4
        x[torch.arange(m, device=dev), torch.randint(0, k, (m,), device=dev)] = 1e4
    elif case == "tiny_and_huge_rows":
        x[: m // 2] *= 1e-30
        x[m // 2:] *= 1e30
    elif case == "zero_rows":
        x[::3] = 0
        w[5] = 0
    b = torch.randn(n, device=dev)
    dy = torch.randn(m, n, device=dev)
    if case == "row_scales":
        dy *= 10.0 ** (torch.rand(m, 1, device=dev) * 12 - 6)
    ref_y = x.double() @ w.double().t()
    den_y = (x.double().abs() @ w.double().abs().t()).clamp_min(1e-300)
    ref_dx = dy.double() @ w.double()
    den_dx = (dy.double().abs() @ w.double().abs()).clamp_min(1e-300)

    def errors(mode, pair):
        prev, was = ops.linear_set_mode(mode), ops.linear_set_pair(pair)
        try:
            served = lib.stemgnn_linear_bigtile_calls()
            y, _, _ = ops.linear_fwd(x, w, None, None, None, False)  # (no bias: its rounding is not the product's error)
            dx = ops.linear_bwd_data(dy, w)
            took = lib.stemgnn_linear_bigtile_calls() - served
        finally:
            ops.linear_set_mode(prev)
            ops.linear_set_pair(was)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(dx).all())
        e_y = float(((y.double() - ref_y).abs() / den_y).max())
        e_dx = float(((dx.double() - ref_dx).abs() / den_dx).max())
        if pair and mode == 1:
            # the format's error model, element by element (csrc/bigtile.hip): 2^-24-level relative terms on
            # sum |x||w| (representation of elements near their row's largest, the dropped lo*lo product, the fp32
            # accumulation of 768 terms: the fp32 chain itself reaches 29 x 2^-24 on the widest-range case, the
            # bf16-piece form 13 x) + 2^-39 of a row's LARGEST magnitude per element of that row (elements more than 2^17 below
            # their row's largest keep an absolute, not a relative, precision)
            xa, wa, ga = x.double().abs(), w.double().abs(), dy.double().abs()
            bound_y = 32 * 2.0 ** -24 * den_y + 4 * 2.0 ** -39 * (xa.sum(1, keepdim=True) * wa.amax(1)[None, :] +
                                                                 xa.amax(1, keepdim=True) * wa.sum(1)[None, :])
            assert bool(((y.double() - ref_y).abs() <= bound_y).all())
            bound_dx = 32 * 2.0 ** -24 * den_dx + 4 * 2.0 ** -39 * (ga.sum(1, keepdim=True) * wa.amax(0)[None, :] +
                                                                   ga.amax(1, keepdim=True) * wa.sum(0)[None, :])
            assert bool(((dx.double() - ref_dx).abs() <= bound_dx).all())
        return e_y, e_dx, took, y

    f32_y, f32_dx, t0, _ = errors(0, 1)        # fp32-MFMA tile kernels (the core does not serve mode 0)
    b3_y, b3_dx, t3, _ = errors(1, 0)          # the core, three bf16 pieces
    p_y, p_dx, t2, y_pair = errors(1, 1)       # the core, pair format
    assert (t0, t3, t2) == (0, 2, 2)
    print(f"{case}: forward fp32-chain {f32_y:.2e} bf16x3 {b3_y:.2e} pair {p_y:.2e}; backward-data {f32_dx:.2e} {b3_dx:.2e} {p_dx:.2e}")
    # measured (in this order: fp32 chain, bf16 pieces, pair; forward / backward-data): normal 4.7e-7 1.5e-7 1.5e-7 /
    # 4.3e-7 1.5e-7 1.3e-7; rows of different magnitudes 4.6e-7 1.4e-7 1.4e-7 / 1.0e-6 3.8e-7 3.8e-7; eight decades
    # inside a row 1.8e-6 7.6e-7 8.2e-7 / 8.0e-7 3.1e-7 3.3e-7
    # ... one outlier of 1e4 among 1e-4 values 3.6e-6 1.5e-6 8.5e-6 (forward).  The last is the format's one weakness and
    # is what the second term of its error model says: where a row's huge element meets a weight that happens to be
    # ~1e-7 -- a handful of the 16.7 M outputs --, that weight, 2^21 below its row's largest, is kept to 2^-39 of the row's
    # largest, and the output's sum |x||w| is small enough for it to show.  Everywhere the bound above holds.
    if case != "outlier":
        assert p_y <= 2 * f32_y and p_dx <= 2 * f32_dx          # no worse than twice the fp32 chain (it is 2-3x better)
        assert p_y <= 1.5 * b3_y + 2e-8 and p_dx <= 1.5 * b3_dx + 2e-8  # and level with the three exact bf16 pieces
    else:
        assert p_y <= 4 * f32_y and p_dx <= 2 * f32_dx
    if case == "zero_rows":  # a zero row times anything is exactly zero; with a bias: exactly the bias
        assert float(y_pair[::3].abs().max()) == 0.0 and float(y_pair[:, 5].abs().max()) == 0.0
        yb, _, _ = ops.linear_fwd(x, w, None, None, b, False)
        assert torch.equal(yb[::3], b.expand(yb[::3].shape)) and torch.equal(yb[:, 5], b[5].expand(m))


@pytest.mark.parametrize("mode,pair", [(1, 1), (1, 0), (2, 1)], ids=["exact-pair", "exact-bf16x3", "bf16"])
def test_bigtile_layer_product_with_bf16_stored_operand_zero_rows_and_row_limit(dev, mode, pair):
    """The layer product as BASELINE config 5 runs it, on the big-tile core in every format: lin_l(agg) + lin_r(h) with
    the aggregate holding its leading `x1_rows` rows only (a sampled batch: row tiles behind them contract over lin_r's
    half alone -- the boundary 4 000 is inside a 256-row tile), `h` STORED as bf16 (feature_kind 1), the BatchNorm column
    sums over all rows, and a row limit on the output (the EMA teacher's last layer: statistics over every row, the
    seed rows' values written) -- against the 128-row tile kernels on the same call and against fp64."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    torch.manual_seed(5)
    m, k1, k2, n, rows, keep = 33_000, 768, 768, 768, 4_000, 1_024
    agg = torch.randn(rows, k1, device=dev)                    # a [x1_rows, K1] buffer: rows past it do not exist
    h = (torch.randn(m, k2, device=dev) * 2).bfloat16().contiguous()
    w1, w2 = torch.randn(n, k1, device=dev) * 0.05, torch.randn(n, k2, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    blocks = int(lib.stemgnn_linear_stats_blocks(m, n))
    prev, was_pair = ops.linear_set_mode(mode), ops.linear_set_pair(pair)
    out = {}
    try:
        ops.linear_scratch(m, k1 + k2, n)
        for core_on in (1, 0):
            was = ops.linear_set_bigtile(core_on)
            served = lib.stemgnn_linear_bigtile_calls()
            y = torch.full((keep, n), float("nan"), device=dev)
            part = torch.zeros(blocks, 2, n, device=dev)
            check(lib.stemgnn_linear_fwd_rows_k(agg.data_ptr(), w1.data_ptr(), k1, h.data_ptr(), 1, w2.data_ptr(), k2,
                                                b.data_ptr(), m, n, y.data_ptr(), part.data_ptr(), None, rows, keep, st))
            assert lib.stemgnn_linear_bigtile_calls() - served == (1 if core_on else 0)
            out[core_on] = (y, part.double().sum(0))
            ops.linear_set_bigtile(was)
    finally:
        ops.linear_set_mode(prev)
        ops.linear_set_pair(was_pair)
    r = (lambda t: t.bfloat16().double()) if mode == 2 else (lambda t: t.double())
    aggf = torch.zeros(m, k1, device=dev, dtype=torch.float64)
    aggf[:rows] = r(agg)
    ref = aggf @ r(w1).t() + h.double() @ r(w2).t() + b.double()
    want = (ref[:keep], torch.stack([ref.sum(0), (ref * ref).sum(0)]))
    for name, got_core, got_tile, w_ in zip(("y[:keep]", "column sums"), out[1], out[0], want):
        tol = 2e-5 * float(w_.abs().max())
        torch.testing.assert_close(got_core.double(), w_, rtol=1e-5, atol=tol, msg=lambda s_: f"{name} (core): {s_}")
        torch.testing.assert_close(got_core.double(), got_tile.double(), rtol=1e-5, atol=tol,
                                   msg=lambda s_: f"{name} (core vs tile kernel): {s_}")


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_bigtile_core_random_shapes(dev, seed):
    """Shapes the fixed cases do not reach, drawn per seed: feature extents that are multiples of 64 but NOT of the
    256-wide tile (partial feature tiles on both sides: clamped reads, masked stores, masked statistics), odd numbers of
    K tiles, row counts with ragged last tiles, a zero-leading-rows boundary anywhere, a random row limit on the output;
    exact mode (pair format for forward / backward-data, bf16 pieces for the weight gradient) and bf16 mode; against fp64
    and against the tile kernels."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    n, k1 = 64 * ri(4, 14), 64 * ri(4, 10)
    k2 = 64 * ri(0, 6) if seed % 2 else 0
    m = max(int(1.05e10 / (2.0 * n * min(n, k1 + k2, k1))) + ri(1, 200), 8192 + ri(0, 300))
    rows = ri(0, m) if k2 else -1
    keep = ri(1, m)
    mode = 1 if seed % 3 else 2
    torch.manual_seed(seed)
    a = torch.randn(m, k1, device=dev)
    w = torch.randn(n, k1, device=dev) * 0.1
    a2 = torch.randn(m, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.1 if k2 else None
    b = torch.randn(n, device=dev)
    if rows >= 0:
        a[rows:] = 0
    dy = torch.randn(m, n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    blocks = int(lib.stemgnn_linear_stats_blocks(m, n))
    prev = ops.linear_set_mode(mode)
    out = {}
    try:
        ops.linear_scratch(m, k1 + k2, n)
        for core_on in (1, 0):
            was = ops.linear_set_bigtile(core_on)
            served = lib.stemgnn_linear_bigtile_calls()
            y = torch.full((keep, n), float("nan"), device=dev)
            part = torch.zeros(blocks, 2, n, device=dev)
            check(lib.stemgnn_linear_fwd_rows_k(a.data_ptr(), w.data_ptr(), k1, None if a2 is None else a2.data_ptr(), 0,
                                                None if w2 is None else w2.data_ptr(), k2, b.data_ptr(), m, n, y.data_ptr(),
                                                part.data_ptr(), None, rows, keep, st))
            dx = ops.linear_bwd_data(dy, w)
            dw, db = ops.linear_bwd_weight(dy, a, True)
            assert (lib.stemgnn_linear_bigtile_calls() - served) == (3 if core_on else 0), (m, n, k1, k2)
            out[core_on] = (y, part.double().sum(0), dx, dw, db)
            ops.linear_set_bigtile(was)
    finally:
        ops.linear_set_mode(prev)
    r = (lambda t: t.bfloat16().double()) if mode == 2 else (lambda t: t.double())
    ref = r(a) @ r(w).t() + (r(a2) @ r(w2).t() if k2 else 0) + b.double()
    want = (ref[:keep], torch.stack([ref.sum(0), (ref * ref).sum(0)]), r(dy) @ r(w), r(dy).t() @ r(a), dy.double().sum(0))
    for name, got_core, got_tile, w_ in zip(("y", "column sums", "dx", "dw", "db"), out[1], out[0], want):
        tol = 2e-5 * float(w_.abs().max())
        torch.testing.assert_close(got_core.double(), w_, rtol=1e-5, atol=tol,
                                   msg=lambda s_: f"{name} (core, m={m} n={n} k={k1}+{k2} rows={rows} keep={keep} mode={mode}): {s_}")
        torch.testing.assert_close(got_core.double(), got_tile.double(), rtol=1e-5, atol=tol,
                                   msg=lambda s_: f"{name} (core vs tile kernel): {s_}")


def test_bigtile_core_without_an_arena_is_a_counted_fallback_and_allocates_nothing(dev):
    """The boundary's rule (entry points never allocate or synchronise) for the big-tile core: its
    scratch is the caller's arena.  Without one a qualifying product runs on the tile kernels and the miss is COUNTED
    (no error is swallowed, none is raised); with one, a whole bf16-mode product sequence is served with the device
    allocator untouched -- torch's allocator statistics see no new segment, the arena's bytes do not move."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    m, k, n = 40_000, 512, 768
    x, w, dy = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.1, torch.randn(m, n, device=dev)
    y = torch.empty(m, n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    prev = ops.linear_set_mode(2)
    try:
        ops.linear_release_scratch()
        missed, served = lib.stemgnn_linear_bigtile_fallbacks(), lib.stemgnn_linear_bigtile_calls()
        check(lib.stemgnn_linear_fwd(x.data_ptr(), w.data_ptr(), k, None, None, 0, None, m, n, y.data_ptr(), None, None, -1, st))
        assert lib.stemgnn_linear_bigtile_fallbacks() == missed + 1 and lib.stemgnn_linear_bigtile_calls() == served
        y_tile = y.clone()
        ops.linear_scratch(m, k, n)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(dev)["num_device_alloc"]
        for _ in range(3):
            check(lib.stemgnn_linear_fwd(x.data_ptr(), w.data_ptr(), k, None, None, 0, None, m, n, y.data_ptr(), None, None, -1, st))
            dx = ops.linear_bwd_data(dy, w)
            dw, db = ops.linear_bwd_weight(dy, x, True)
        torch.cuda.synchronize()
        assert lib.stemgnn_linear_bigtile_calls() == served + 9 and lib.stemgnn_linear_bigtile_fallbacks() == missed + 1
        # outputs are torch allocations from its cached pool; the LIBRARY made none: no new device segment appeared
        assert torch.cuda.memory_stats(dev)["num_device_alloc"] == before
        torch.testing.assert_close(y, y_tile, rtol=1e-5, atol=2e-5 * float(y_tile.abs().max()))
    finally:
        ops.linear_set_mode(prev)


def test_bf16_mode_steps_on_the_bigtile_core_allocate_nothing_after_the_first(dev):
    """Round-3 review, item 5, at step level: whole pretraining steps in the bf16 GEMM mode at a width the big-tile core
    takes (N = 50 000, D = 512, H = 4, K = 512: the projections, their backward products AND the large-codebook
    assignment) with the scratch arena the host registered.  After the first step (which sizes the arena and torch's
    pool) two more steps are served by the core -- its counter moves, its fallback counter does not -- while neither
    torch's allocator (no new device segment) nor the device's free memory (a library-side hipMalloc would show there)
    moves."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure
    from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step
    N, E, D, K = 50_000, 400_000, 512, 512
    g = make_graph(N, E, D, 4, kind="U", device=dev)
    params = default_params()
    params.update(input_dim=D, hidden_dim=D, code_dim=D, codebook_size=K, pretrain_batch_size=N)
    torch.manual_seed(0)
    prev = ops.linear_set_mode(2)
    try:
        model = build_model(params, dev).train()
        opt, sched = build_optimizer(model, params)
        gs = GraphStructure(g.edge_index, N, g.xe, validate=True).ensure_transpose()
        ea = EdgeTypeAttr(g.edge_text_feat, g.xe)
        ops.manual_seed(3)
        losses = []
        loss, _, _ = pretrain_step(model, opt, sched, params, g.node_text_feat, gs, ea, N, record_draws=False)
        losses.append(float(loss))
        torch.cuda.synchronize()
        served, missed = lib.stemgnn_linear_bigtile_calls(), lib.stemgnn_linear_bigtile_fallbacks()
        segs, free0 = torch.cuda.memory_stats(dev)["num_device_alloc"], torch.cuda.mem_get_info(dev)[0]
        for _ in range(2):
            loss, _, _ = pretrain_step(model, opt, sched, params, g.node_text_feat, gs, ea, N, record_draws=False)
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert lib.stemgnn_linear_bigtile_calls() - served >= 2 * 10, "the step's large products must run on the core"
        assert lib.stemgnn_linear_bigtile_fallbacks() == missed
        assert int(lib.stemgnn_vq_assign_last_path()) in (0, 4)  # (the assignment runs on the autograd-free forward thread)
        assert torch.cuda.memory_stats(dev)["num_device_alloc"] == segs
        assert torch.cuda.mem_get_info(dev)[0] == free0
        assert all(l == l and abs(l) < 1e6 for l in losses)
    finally:
        ops.linear_set_mode(prev)


def test_deterministic_mode_makes_steps_bit_reproducible(dev):
    """stemgnn_set_deterministic(1): the decoders' backward scatters add in a fixed order (edges grouped by node)
    instead of with fp32 atomics -- the only order-dependent arithmetic on the path.  Two runs of three optimiser steps
    from the same state and draws end in identical parameter bits, and agree with the default (atomic) mode to
    rounding."""
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib
    from stem_gnn_amd.data.sampler import HipNeighborSampler
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import pretrain_step, default_params
    D, L, H, K = 64, 2, 4, 64
    g = make_graph(5000, 60000, D, 4, kind="U", device=dev)
    s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [6, 6], seed=2)
    b = s.sample(torch.randperm(5000, device=dev)[:96])
    x = g.node_text_feat[b.n_id]
    params = default_params()

    def run(det, steps):
        prev = lib.stemgnn_set_deterministic(det)
        try:
            _, gm = make_models(D, L, H, K, D, dev)
            opt = torch.optim.AdamW(gm.parameters(), lr=1e-3, weight_decay=1e-5)
            ops.manual_seed(33)
            for _ in range(steps):
                pretrain_step(gm, opt, None, params, x, b.graph, EdgeTypeAttr(g.edge_text_feat, b.xe), 96)
            return ([p.detach().clone() for p in gm.parameters()],
                    [None if p.grad is None else p.grad.detach().clone() for p in gm.parameters()])
        finally:
            lib.stemgnn_set_deterministic(prev)

    (pa, ga), (pc, gc) = run(1, 3), run(1, 3)
    for p1, p2 in zip(pa, pc):
        assert torch.equal(p1, p2)
    for g1, g2 in zip(ga, gc):
        assert (g1 is None and g2 is None) or torch.equal(g1, g2)
    # one step from identical parameters: the two modes' gradients differ by summation order only (AdamW would turn a
    # last-bit difference of a near-zero gradient into a full step, so parameters are not compared across modes)
    (_, gd), (_, ge) = run(1, 1), run(0, 1)
    seen = 0
    for g1, g2 in zip(gd, ge):
        if g1 is None:
            continue
        seen += 1
        # (a bias in front of a BatchNorm has a mathematically zero gradient: what is compared there is rounding noise)
        torch.testing.assert_close(g1, g2, rtol=1e-3, atol=max(1e-5 * float(g2.abs().max()), 1e-6))
    assert seen > 10


def test_encoder_five_layers_queue_more_weight_gradients_than_a_batch_holds(dev):
    """Five SAGE layers queue ten weight-gradient products in the encoder's backward: more than one DwBatch table (8),
    so the batch flushes itself part-way.  Gradients against the CPU oracle as in test_encoder_fwd_bwd_vs_oracle."""
    from stem_gnn_amd import ops
    N, E, D, L = 700, 6000, 32, 5
    om, gm = make_models(D, L, 2, 16, D, dev)
    torch.manual_seed(9)
    x = torch.randn(N, D)
    ei = torch.randint(0, N, (2, E))
    oe, ge = om.encoder, gm.encoder
    oe.train(); ge.train()
    xg = x.to(dev).requires_grad_(True)
    zg = ge(xg, ei.to(dev), None)
    masks = [ops.dropout_keep_mask(N * D, 0.15, s, o, dev).view(N, D).cpu() for (s, o) in ge.last_dropout_keys]
    xr = x.clone().requires_grad_(True)
    zr = oe(xr, ei, None, dropout_masks=masks)
    torch.testing.assert_close(zg.detach().cpu(), zr.detach(), rtol=1e-4, atol=1e-4)
    w = torch.randn(N, D)
    (zr * w).sum().backward()
    (zg * w.to(dev)).sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-3, atol=2e-4)
    for (n1, p1), (n2, p2) in zip(oe.named_parameters(), ge.named_parameters()):
        assert n1 == n2
        torch.testing.assert_close(p2.grad.cpu(), p1.grad, rtol=2e-3, atol=5e-4, msg=lambda m: f"{n1}: {m}")


def test_loss_curve_one_scheduler_period(dev, capsys):
    """VERDICT round 2, item 2 / north_star "loss curve matching reference to 1e-4": 60 optimiser steps with the
    reference's scheduler (utils/others.py:138-145 stepped per BATCH, pretrain.py:64-65: lr reaches 0 at step 50 and
    climbs back), the same graph with fresh draws every step, every loss term of every step against the CPU oracle
    replaying the HIP run's draws: 1e-4 relative (+1e-5 absolute).  Near-ties of the code assignment (top-2 similarity gap
    <= 1e-5, decided by fp32 summation order alone) are replayed like any other draw and counted; an assignment that
    differs beyond a near-tie fails inside the oracle step.  Prints the per-term maximum deviation (DESIGN.md section 4
    quotes it) and, per parameter, how far the two runs' parameters are apart after 60 steps."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.graph import EdgeTypeAttr
    from stem_gnn_amd.pretrain import pretrain_step, default_params
    from stem_gnn_amd.utils.others import get_scheduler
    N, E, D, L, H, K = 600, 5000, 64, 2, 4, 64
    bs, steps = 200, 60
    om, gm = make_models(D, L, H, K, D, dev)
    params = default_params()
    torch.manual_seed(12)
    x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1)
    half = torch.randint(0, N, (2, E // 2))
    ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
    table = torch.nn.functional.normalize(torch.randn(4, D), dim=-1)
    et = torch.randint(0, 4, (E,))
    opt_o = torch.optim.AdamW(om.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    opt_g = ops.FusedAdamW(gm.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    sch_o, sch_g = get_scheduler(opt_o, True, 50), get_scheduler(opt_g, True, 50)
    xg, eig = x.to(dev), ei.to(dev)
    eag = EdgeTypeAttr(table.to(dev), et.to(dev))
    ops.manual_seed(123)
    worst, where, ties, lrs = {}, {}, 0, []
    failures = []
    for step in range(steps):
        lrs.append(opt_g.param_groups[0]["lr"])
        loss_g, losses_g, draws = pretrain_step(gm, opt_g, sch_g, params, xg, eig, eag, bs)
        cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
        loss_o, losses_o, _ = O.pretrain_step(om, opt_o, sch_o, params, x, ei, table[et], bs, cpu_draws)
        ties += om.vq.last_tie_adopted
        terms = dict(losses_o)
        terms["total"] = loss_o
        got = {k: v for k, v in losses_g.items()}
        got["total"] = loss_g
        for k, vo in terms.items():
            a, b = float(got[k].reshape(-1)[0]), float(vo.reshape(-1)[0])
            dev_rel = abs(a - b) / max(abs(b), 1e-12)
            if dev_rel > worst.get(k, -1.0):
                worst[k], where[k] = dev_rel, step
            if abs(a - b) > 1e-4 * abs(b) + 1e-5:
                failures.append((step, k, a, b))
    assert abs(lrs[50]) < 1e-12 and lrs[25] == pytest.approx(0.5e-4, rel=1e-6) and lrs[59] > lrs[51] > 0  # the period
    with capsys.disabled():
        print(f"\\nloss-curve replay, {steps} steps, N={N} D={D}: near-tie assignments replayed: {ties}")
        for k in worst:
            print(f"  {k:22s} max relative deviation {worst[k]:.3e} (step {where[k]})")
        for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
            d = float((p2.detach().cpu() - p1.detach()).abs().max())
            print(f"  param {n1:40s} max |diff| {d:.3e}")
    assert not failures, failures[:10]
    # lin_l.bias sits in front of BatchNorm, which subtracts the column mean: the losses do not depend on it at all, its
    # true gradient is exactly zero, and what reaches Adam on either side is rounding noise that Adam's normalisation
    # turns into +-lr steps.  SHOWN here, not assumed: shifting every lin_l.bias by 0.5 leaves every loss term where
    # it was (the oracle, same draws), while the same shift of a BatchNorm bias moves them.
    import copy

    def losses_with_shift(pattern):
        m = copy.deepcopy(om)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if n_.startswith("encoder.") and n_.endswith(pattern):
                    p_.add_(0.5)
        sgd = torch.optim.SGD(m.parameters(), lr=0.0)
        replay = {k_: v_ for k_, v_ in cpu_draws.items() if k_ != "vq_indices"}  # the oracle's own arg-max here
        return O.pretrain_step(m, sgd, None, params, x, ei, table[et], bs, replay)[1]

    base, shifted, moved = losses_with_shift("<none>"), losses_with_shift("lin_l.bias"), losses_with_shift("norms.0.bias")
    for k in base:
        torch.testing.assert_close(shifted[k], base[k], rtol=2e-6, atol=1e-7, msg=lambda m: f"lin_l.bias shift moved {k}: {m}")
    assert any(abs(float(moved[k].reshape(-1)[0]) - float(base[k].reshape(-1)[0])) > 1e-3 * abs(float(base[k].reshape(-1)[0]))
               for k in ("feat_recon_loss", "topo_sem_recon_loss", "sem_recon_loss"))
    for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()):
        if n1.endswith("lin_l.bias"):
            continue  # see above
        torch.testing.assert_close(p2.detach().cpu(), p1.detach(), rtol=1e-3, atol=1e-4, msg=lambda m: f"{n1}: {m}")
