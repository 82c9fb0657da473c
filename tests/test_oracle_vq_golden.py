"""Pin the CPU oracle's vector quantiser to the reference: compare
oracle/stem_oracle.OracleVectorQuantize with the golden vectors that
tests/golden/gen_vq_golden.py produced by importing the reference's model/vq.py."""
import glob
import os

import pytest
import torch

from oracle.stem_oracle import OracleVectorQuantize

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vq_*.pt")))


def build_from_fixture(fx):
    N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
    vq = OracleVectorQuantize(D, K, Dc, H, decay=0.8, commitment_weight=10.0, orthogonal_reg_weight=1.0,
                              orthogonal_reg_max_codes=ortho_max, ema_update=bool(ema))
    state = {k[len("state0."):]: v for k, v in fx.items() if k.startswith("state0.")}
    vq.load_state_dict(state)  # exact key/shape contract of the reference
    return vq


def test_fixtures_present():
    assert len(FIXTURES) == 18


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_vq_matches_reference(path):
    fx = torch.load(path, weights_only=True)
    vq = build_from_fixture(fx)
    vq.train()
    z = fx["z"].clone().requires_grad_(True)
    q, ind, loss, oq = vq(z, ortho_ids=fx["ortho_ids"])
    assert ind.dtype == torch.int64
    assert torch.equal(ind, fx["train.embed_ind"])  # same torch einsum on the same box -> bit-exact
    torch.testing.assert_close(q, fx["train.quantize"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(oq, fx["train.orig_quantize"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, fx["train.loss"], rtol=1e-5, atol=1e-6)
    w = torch.linspace(-1.0, 1.0, q.numel()).view_as(q)
    (loss.sum() + (q * w).sum()).backward()
    torch.testing.assert_close(z.grad, fx["train.grad_z"], rtol=1e-4, atol=1e-6)
    for pn, p in vq.named_parameters():
        key = "train.grad." + pn
        if key in fx:
            torch.testing.assert_close(p.grad, fx[key], rtol=1e-4, atol=1e-6)
        else:
            assert p.grad is None
    if int(fx["meta"][6]):
        for k, v in vq.state_dict().items():
            if k.startswith("_codebook."):
                torch.testing.assert_close(v, fx["post." + k], rtol=1e-5, atol=1e-6)
    # eval mode on the initial state
    vq2 = build_from_fixture(fx)
    vq2.eval()
    with torch.no_grad():
        q2, ind2, loss2, oq2 = vq2(fx["z"])
    assert torch.equal(ind2, fx["eval.embed_ind"])
    torch.testing.assert_close(q2, fx["eval.quantize"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(oq2, fx["eval.orig_quantize"], rtol=1e-5, atol=1e-6)
    assert float(loss2) == 0.0


# ---- production-shape fixtures (round 3): inputs regenerated from seeds, results stored as row checks ---------------
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import prod_recipe as R  # noqa: E402

PROD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "prod_vq_*.pt")))


def load_prod(path):
    """(fixture, regenerated state, regenerated input); the regenerated tensors are checked against the fixture's
    checksums first, so a torch build whose CPU generator differs fails HERE and not as a parity error."""
    fx = torch.load(path, weights_only=True)
    N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
    state = R.make_state(D, H, K, Dc, seed)
    z = R.make_input(N, D, seed)
    for k, v in state.items():
        torch.testing.assert_close(R.checksum(v), fx["chk.state0." + k], rtol=1e-12, atol=1e-9, msg=f"recipe drift: {k}")
    torch.testing.assert_close(R.checksum(z), fx["chk.z"], rtol=1e-12, atol=1e-9, msg="recipe drift: z")
    return fx, state, z


def test_prod_fixtures_present():
    assert [os.path.basename(p) for p in PROD] == ["prod_vq_big_s0.pt", "prod_vq_k512_s0.pt", "prod_vq_k512_s1.pt", "prod_vq_ws_s0.pt"]
    assert all(os.path.getsize(p) < (1 << 20) for p in PROD)


@pytest.mark.parametrize("path", PROD, ids=[os.path.basename(p) for p in PROD])
def test_oracle_vq_matches_reference_at_production_shapes(path):
    """(N=1000, D=128, H=4, K=512, Dc=128), and one case past the row gate of the
    weight-stationary assignment kernel (N=16 500, K=Dc=128): the oracle against what the reference's vq.py returned."""
    fx, state, z = load_prod(path)
    N, D, H, K, Dc, ortho_max, ema, seed = fx["meta"].tolist()
    vq = OracleVectorQuantize(D, K, Dc, H, decay=0.8, commitment_weight=10.0, orthogonal_reg_weight=1.0,
                              orthogonal_reg_max_codes=ortho_max, ema_update=False)
    vq.load_state_dict(state)
    vq.train()
    zt = z.clone().requires_grad_(True)
    q, ind, loss, oq = vq(zt, ortho_ids=fx["ortho_ids"])
    assert torch.equal(ind, fx["train.embed_ind"].long())  # same torch einsum on the same box -> bit-exact
    torch.testing.assert_close(loss, fx["train.loss"], rtol=1e-5, atol=1e-6)
    R.assert_rows_close(q.detach(), fx["train.quantize.rows"], rtol=1e-5, what="quantize")
    R.assert_rows_close(oq.detach(), fx["train.orig_quantize.rows"], rtol=1e-5, what="orig_quantize")
    (loss.sum() + (q * R.make_upstream(N, D)).sum()).backward()
    R.assert_rows_close(zt.grad, fx["train.grad_z.rows"], rtol=1e-4, what="grad_z")
    for pn, p in vq.named_parameters():
        key = "train.grad." + pn + ".rows"
        if key in fx:
            R.assert_rows_close(p.grad if p.grad.dim() > 1 else p.grad.view(1, -1), fx[key], rtol=1e-4, what=pn)
        else:
            assert p.grad is None
    if "eval.quantize.rows" in fx:
        vq.eval()
        with torch.no_grad():
            q2, ind2, loss2, _ = vq(z)
        assert torch.equal(ind2, fx["train.embed_ind"].long()) and float(loss2) == 0.0
        R.assert_rows_close(q2, fx["eval.quantize.rows"], rtol=1e-5, what="eval quantize")
