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
