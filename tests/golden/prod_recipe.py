"""Seed-reproducible inputs of the production-shape quantiser fixtures (``prod_vq_*.pt``).

At the shapes the benchmark's kernels run (D = 128, H = 4, Dc = 128, K = 512 or 128, N up to 16 500) the initial state
and the input are 1 - 8 MB, too large to commit per fixture.  They are regenerated instead -- by this file, on the CPU,
from the seeds the fixture stores -- both by the generator (tests/golden/gen_vq_prod_golden.py, which feeds them to the
reference's VectorQuantize) and by the tests.  torch's CPU generator is deterministic for a given torch build; the
fixture also stores a checksum of every regenerated tensor, so a mismatch is reported as such, not as a parity failure.
"""
import math

import torch

CASES = {
    # name: (N, D, H, K, Dc, ortho_max); SURVEY.md section 8(c) lists (1000, 128, 4, 512, 128); "ws" crosses the row gate
    # of the weight-stationary assignment kernel (csrc/wsgemm.hip: K = Dc = 128, N >= 16 384) with a ragged last tile
    "k512": (1000, 128, 4, 512, 128, 32),
    "ws": (16500, 128, 4, 128, 128, 32),
    # round 4: reaches the large-codebook assignment (K >= 512 and Dc >= 256 with >= 8 192 rows: BASELINE configs 3 / 5
    # take that path), ragged in its last 128-row and 256-row tiles
    "big": (9000, 256, 2, 512, 256, 32),
}


def make_state(D, H, K, Dc, seed):
    """The reference's state-dict keys and shapes with nn.Linear-like / unit-norm values."""
    g = torch.Generator().manual_seed(50_000 + seed)
    HD = H * Dc

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    embed = torch.nn.functional.normalize(torch.randn(H, K, Dc, generator=g), dim=-1)
    return {
        "project_in.weight": uni((HD, D), D), "project_in.bias": uni((HD,), D),
        "project_out.weight": uni((D, HD), HD), "project_out.bias": uni((D,), HD),
        "_codebook.initted": torch.ones(1), "_codebook.cluster_size": torch.zeros(H, K),
        "_codebook.embed_avg": embed.clone(), "_codebook.embed": embed,
    }


def make_input(N, D, seed):
    g = torch.Generator().manual_seed(60_000 + seed)
    return torch.randn(N, D, generator=g) * 1.5


def make_upstream(N, D):
    """Fixed upstream gradient of ``quantize`` (a function of the position only)."""
    return torch.sin(torch.arange(N * D, dtype=torch.float64) * 0.37).float().view(N, D)


def checksum(t):
    """[sum, sum of |.|] in fp64 -- identifies a regenerated tensor."""
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum()])


def row_checks(t):
    """Per row of a 2-D view of ``t`` (last dimension = columns): [sum, dot with a fixed vector, sum of |.|], fp64
    arithmetic stored as fp32.  Three numbers per row stand in for the row in a fixture that must stay small."""
    m = t.reshape(-1, t.shape[-1]).double()
    v = torch.cos(torch.arange(m.size(1), dtype=torch.float64) * 0.61 + 0.3)
    return torch.stack([m.sum(1), m @ v, m.abs().sum(1)], dim=1).float()


def assert_rows_close(got, want_checks, rows_ok=None, rtol=1e-4, what=""):
    """``row_checks(got)`` against the stored checks: the sum and the dot product may deviate by rtol times the row's
    L1 norm (what a relative error of rtol per element can move them by)."""
    gc = row_checks(got)
    if rows_ok is not None:
        gc, want_checks = gc[rows_ok], want_checks[rows_ok]
    scale = want_checks[:, 2:3].clamp_min(1e-6)
    err = ((gc[:, :2] - want_checks[:, :2]).abs() / scale).max()
    l1 = ((gc[:, 2] - want_checks[:, 2]).abs() / scale[:, 0]).max()
    assert float(err) <= rtol and float(l1) <= rtol, f"{what}: row checks off by {float(err):.3g} / {float(l1):.3g} of the row's L1 norm"
