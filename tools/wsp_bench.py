"""Weight-stationary products of the D = 128 configurations in the pair format (csrc/wspair.hip) against the bf16-piece
kernels (csrc/wsgemm.hip, csrc/linear.hip) at C4's batch size: time (HIP events, back to back on random data) and the
largest error against fp64 relative to the result's largest magnitude.

    python tools/wsp_bench.py [--rows 101927] [--head 11200] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=101927)
    ap.add_argument("--head", type=int, default=11200)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    dev = torch.device("cuda:0")
    M, H = args.rows, args.head
    torch.manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    h = torch.randn(M, 128, device=dev) * (0.2 + 2 * torch.rand(M, 1, device=dev))
    agg = torch.randn(H, 128, device=dev)
    cases = [("lin_r 128->128 + stats", None, 128, -1, M, True), ("layer (head rows) + stats", agg, 128, H, M, True),
             ("layer, seed rows stored + stats", agg, 128, H, 1024, True), ("project_in 128->512", None, 512, -1, M, False)]
    for name, x1, n, rows, keep, stats in cases:
        w2 = torch.randn(n, 128, device=dev) * 0.1
        w1 = torch.randn(n, 128, device=dev) * 0.1
        b = torch.randn(n, device=dev)
        blocks = max(int(lib.stemgnn_linear_stats_blocks(M, n)), 1)
        y = torch.empty(keep, n, device=dev)
        part = torch.zeros(blocks, 2, n, device=dev)

        def fwd():
            if x1 is None:
                check(lib.stemgnn_linear_fwd_rows_k(h.data_ptr(), w2.data_ptr(), 128, None, 0, None, 0, b.data_ptr(), M, n,
                                                    y.data_ptr(), part.data_ptr() if stats else None, None, -1, keep, st))
            else:
                check(lib.stemgnn_linear_fwd_rows_k(x1.data_ptr(), w1.data_ptr(), 128, h.data_ptr(), 0, w2.data_ptr(), 128,
                                                    b.data_ptr(), M, n, y.data_ptr(), part.data_ptr() if stats else None, None,
                                                    rows, keep, st))
        ref = h.double() @ w2.double().t() + b.double()
        if x1 is not None:
            ref[:rows] += x1.double() @ w1.double().t()
        out = []
        for pair in (1, 0):
            was = ops.linear_set_pair(pair)
            us = timeit(fwd, args.reps)
            err = float((y.double() - ref[:keep]).abs().max() / ref.abs().max())
            serr = float((part.double().sum(0)[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max()) if stats else 0.0
            ops.linear_set_pair(was)
            out.append((us, err, serr))
        print(f"{name:34s} pair {out[0][0]:7.1f} us (err {out[0][1]:.1e}, sums {out[0][2]:.1e})   "
              f"bf16 pieces {out[1][0]:7.1f} us (err {out[1][1]:.1e}, sums {out[1][2]:.1e})   x{out[1][0] / out[0][0]:.2f}", flush=True)
    # backward-data: dx = dy w
    dy = torch.randn(M, 128, device=dev) * (0.2 + 2 * torch.rand(M, 1, device=dev))
    for n in (128, 512):
        w = torch.randn(128, n, device=dev) * 0.1  # [N_out = 128][K_in = n]: dx [M, n] = dy [M, 128] w
        ref = dy.double() @ w.double()
        out = []
        for pair in (1, 0):
            was = ops.linear_set_pair(pair)
            us = timeit(lambda: ops.linear_bwd_data(dy, w), args.reps)
            err = float((ops.linear_bwd_data(dy, w).double() - ref).abs().max() / ref.abs().max())
            ops.linear_set_pair(was)
            out.append((us, err))
        print(f"{'backward-data 128 -> ' + str(n):34s} pair {out[0][0]:7.1f} us (err {out[0][1]:.1e})   "
              f"bf16 pieces {out[1][0]:7.1f} us (err {out[1][1]:.1e})   x{out[1][0] / out[0][0]:.2f}", flush=True)


if __name__ == "__main__":
    main()
