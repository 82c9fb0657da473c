"""Per-kernel micro-benchmarks on C4-batch shapes (run on the GPU box):
python tools/kbench.py [names...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops  # noqa: E402
from stem_gnn_amd.graph import GraphStructure, EdgeTypeAttr  # noqa: E402

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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    which = set(sys.argv[1:])
    M, D, H, K = 102400, 128, 4, 128
    torch.manual_seed(0)
    x = torch.randn(M, D, device=dev)
    if not which or "gemm" in which:
        for lib in ("default", "hipblaslt", "cublas"):
            if lib != "default":
                try:
                    torch.backends.cuda.preferred_blas_library(lib)
                except Exception as e:  # noqa
                    print("preferred_blas_library", lib, "failed", e)
                    continue
            for (m, k, n) in [(M, 128, 128), (M, 256, 128), (M, 128, 512), (M, 512, 128)]:
                a = torch.randn(m, k, device=dev)
                w = torch.randn(n, k, device=dev)
                b = torch.randn(n, device=dev)
                us = timeit(lambda: torch.nn.functional.linear(a, w, b))
                g = torch.randn(m, n, device=dev)
                us_dx = timeit(lambda: g @ w)
                us_dw = timeit(lambda: g.t() @ a)
                fl = 2.0 * m * k * n
                print(f"[{lib}] linear {m}x{k}->{n}: fwd {us:8.1f} us ({fl/us/1e6:6.1f} TF/s)  dX {us_dx:8.1f} us ({fl/us_dx/1e6:6.1f})  dW {us_dw:8.1f} us ({fl/us_dw/1e6:6.1f})")
    if not which or "linear" in which:
        for (m, k1, k2, n) in [(M, 128, 0, 128), (M, 128, 128, 128), (M, 128, 0, 512), (M, 512, 0, 128)]:
            a = torch.randn(m, k1, device=dev)
            w = torch.randn(n, k1, device=dev)
            a2 = torch.randn(m, k2, device=dev) if k2 else None
            w2 = torch.randn(n, k2, device=dev) if k2 else None
            b = torch.randn(n, device=dev)
            us = timeit(lambda: ops.linear_fwd(a, w, a2, w2, b, False))
            us_s = timeit(lambda: ops.linear_fwd(a, w, a2, w2, b, True))
            g = torch.randn(m, n, device=dev)
            us_dw = timeit(lambda: ops.linear_bwd_weight(g, a, True))
            wt = ops.transpose(w)
            us_dx = timeit(lambda: ops.linear_fwd(g, wt, None, None, None))
            fl = 2.0 * m * (k1 + k2) * n
            fl1 = 2.0 * m * k1 * n
            print(f"[hip] linear {m}x({k1}+{k2})->{n}: fwd {us:8.1f} us ({fl/us/1e6:6.1f} TF/s) +stats {us_s:8.1f}  dX {us_dx:8.1f} us ({fl1/us_dx/1e6:6.1f})  dW {us_dw:8.1f} us ({fl1/us_dw/1e6:6.1f})")
    if not which or "vq" in which:
        for (k_, dc) in [(128, 128), (512, 128)]:
            xp = torch.randn(M, H * dc, device=dev)
            emb = torch.nn.functional.normalize(torch.randn(H, k_, dc, device=dev), dim=-1)
            us = timeit(lambda: ops.VqAssignFn.apply(xp, emb, H, True))
            fl = 2.0 * M * H * k_ * dc
            print(f"vq_assign N={M} H={H} K={k_} Dc={dc}: {us:8.1f} us ({fl/us/1e6:6.1f} TF/s, {3*M*H*dc*4/us/1e3:6.1f} GB/s io)")
    if not which or "bn" in which:
        y = torch.randn(M, D, device=dev)
        gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        rm, rv = torch.zeros(D, device=dev), torch.ones(D, device=dev)
        us = timeit(lambda: ops.bn_stats(y, 1e-5, rm, rv, 0.1))
        print(f"bn_stats {M}x{D}: {us:8.1f} us ({M*D*4/us/1e3:6.1f} GB/s)")
        yr = y.clone().requires_grad_(True)
        def fb():
            out = ops.BnActDropFn.apply(yr, gam.requires_grad_(True), bet.requires_grad_(True), rm, rv, True, 0.1, 1e-5, 1, 0.0, 0.15, 1, 2)
            out.backward(y)
        us = timeit(fb)
        print(f"bn_act_drop fwd+bwd {M}x{D}: {us:8.1f} us")
    if not which or "agg" in which:
        for (n, e) in [(102400, 112000), (100000, 1000000)]:
            ei = torch.randint(0, n, (2, e), device=dev)
            if n == 102400:  # C4-like: only the first 11k nodes receive edges
                ei[1] = torch.randint(0, 11264, (e,), device=dev)
            xx = torch.randn(n, D, device=dev, requires_grad=True)
            et = torch.randint(0, 4, (e,), device=dev)
            tab = torch.randn(4, D, device=dev)
            g = GraphStructure(ei, n, et)
            g.ensure_transpose()
            us = timeit(lambda: ops.sage_agg_fwd(xx.detach(), g, None, tab))
            by = ops.k1_algorithmic_bytes(n, e, D, "table", 4)
            ga = torch.randn(n, D, device=dev)
            us_b = timeit(lambda: ops.sage_agg_bwd(ga, xx.detach(), g, None, tab))
            ead = tab[et]
            us_d = timeit(lambda: ops.sage_agg_fwd(xx.detach(), g, ead, None))
            byd = ops.k1_algorithmic_bytes(n, e, D, "dense")
            us_csr = timeit(lambda: GraphStructure(ei, n, et, validate=False))
            print(f"K1 fwd N={n} E={e}: table {us:7.1f} us ({by/us/1e3:7.1f} GB/s)  dense {us_d:7.1f} us ({byd/us_d/1e3:7.1f} GB/s)  K2 bwd {us_b:7.1f} us  csr_build {us_csr:7.1f} us")


if __name__ == "__main__":
    main()
