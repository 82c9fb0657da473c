# Round-2 record: ran against the library of commit b742edb, which still exported stemgnn_pgemm_* (tools/micro/pgemm.hip).
"""Plane-operand products (csrc/pgemm.hip) beside the register-staged kernels (csrc/linear.hip): correctness
(bit-identical results expected) and time per launch at C4-batch shapes.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M = int(os.environ.get("M", 102400))
    torch.manual_seed(0)
    for (k1, k2, n, rows) in [(128, 0, 128, -1), (128, 128, 128, -1), (128, 128, 128, 11264), (128, 0, 512, -1),
                              (512, 0, 128, -1)]:
        a = torch.randn(M, k1, device=dev) * (1 + 3 * torch.rand(M, 1, device=dev))
        w = torch.randn(n, k1, device=dev) * 0.1
        a2 = torch.randn(M, k2, device=dev) if k2 else None
        w2 = torch.randn(n, k2, device=dev) * 0.1 if k2 else None
        b = torch.randn(n, device=dev)
        if rows >= 0:
            a[rows:] = 0
        y0, p0, _ = ops.linear_fwd(a, w, a2, w2, b, True, rows)
        ap = ops.split_planes(a)
        a2p = ops.split_planes(a2) if k2 else None
        wp = ops.weight_planes([w] + ([w2] if k2 else []), [False] * (2 if k2 else 1))
        y1, p1 = ops.pgemm_fwd(ap, wp[0], a2p, wp[1] if k2 else None, b, True, rows)
        torch.cuda.synchronize()
        err = (y1 - y0).abs().max().item()
        s0 = p0.sum(0)
        s1 = p1.sum(0)
        serr = ((s1 - s0).abs() / (s0.abs() + 1)).max().item()
        t0 = timeit(lambda: ops.linear_fwd(a, w, a2, w2, b, False, rows))
        t1 = timeit(lambda: ops.pgemm_fwd(ap, wp[0], a2p, wp[1] if k2 else None, b, False, rows))
        t1s = timeit(lambda: ops.pgemm_fwd(ap, wp[0], a2p, wp[1] if k2 else None, b, True, rows))
        tsp = timeit(lambda: ops.split_planes(a))
        print(f"fwd M={M} K={k1}+{k2} N={n} rows={rows}: staged {t0:7.1f} us  planes {t1:7.1f} us (+stats {t1s:7.1f})  "
              f"split pass {tsp:6.1f} us  max|diff| {err:.3g}  stats rel diff {serr:.2g}")
    # backward-data through transposed weight planes: dx = dy w
    for (n, k) in [(128, 128), (128, 512), (512, 128)]:
        dy = torch.randn(M, n, device=dev)
        w = torch.randn(n, k, device=dev) * 0.1
        d0 = ops.linear_bwd_data(dy, w)
        wt = ops.weight_planes([w], [True])[0]
        dyp = ops.split_planes(dy)
        d1, _ = ops.pgemm_fwd(dyp, wt)
        torch.cuda.synchronize()
        err = (d1 - d0).abs().max().item()
        t0 = timeit(lambda: ops.linear_bwd_data(dy, w))
        t1 = timeit(lambda: ops.pgemm_fwd(dyp, wt))
        print(f"bwd-data M={M} N={n} -> K={k}: staged {t0:7.1f} us  planes {t1:7.1f} us  max|diff| {err:.3g}")
    # weight gradient
    for (n, k, rows) in [(128, 128, -1), (128, 128, 11264), (512, 128, -1), (128, 512, -1)]:
        dy = torch.randn(M, n, device=dev)
        x = torch.randn(M, k, device=dev)
        m = M if rows < 0 else rows
        w0, b0 = ops.linear_bwd_weight(dy[:m], x[:m], True)
        dyp, xp = ops.split_planes(dy), ops.split_planes(x)
        w1, b1 = ops.pgemm_dw(dyp, xp, True, rows)
        torch.cuda.synchronize()
        err = (w1 - w0).abs().max().item()
        berr = (b1 - b0).abs().max().item()
        t0 = timeit(lambda: ops.linear_bwd_weight(dy[:m], x[:m], True))
        t1 = timeit(lambda: ops.pgemm_dw(dyp, xp, True, rows))
        print(f"dW M={m} N={n} K={k}: staged {t0:7.1f} us  planes {t1:7.1f} us  max|diff| {err:.3g} (|dw| max {w0.abs().max().item():.3g})  "
              f"db diff {berr:.3g}")


if __name__ == "__main__":
    main()
