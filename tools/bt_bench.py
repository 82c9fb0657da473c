"""Big-tile core (csrc/bigtile.hip) against the 128-row tile kernels at the D = 768 shapes of BASELINE configs 3 / 5:
forward, backward-data and weight gradient of project_in (M x 768 x 3072) and of the layer product (M x 1536 x 768), and
the code assignment (K = 512 and 2048, Dc = 768), each timed back to back with HIP events on random data.

    python tools/bt_bench.py [--rows 169343] [--reps 5]

Prints per product: time, fp32-equivalent TFLOP/s (2 M N K / t) and, for the core, EXECUTED bf16 TFLOP/s (x 6 in the exact
mode, x 1 in the bf16 mode; the cut passes are inside the time)."""
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
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=169343)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--modes", default="1,2")
    ap.add_argument("--pair", type=int, default=1, help="exact mode: 1 = pair format (default), 0 = three bf16 pieces")
    args = ap.parse_args()
    from stem_gnn_amd import ops
    from stem_gnn_amd._lib import lib, check
    dev = torch.device("cuda:0")
    ops.linear_set_pair(args.pair)
    M = args.rows
    torch.manual_seed(0)
    shapes = [("project_in 768->3072", 768, 0, 3072), ("layer 768+768->768", 768, 768, 768), ("lin 768->768", 768, 0, 768)]
    for mode in [int(v) for v in args.modes.split(",")]:
        prev = ops.linear_set_mode(mode)
        pieces = (3 if ops.linear_set_pair(-1) else 6) if mode == 1 else 1  # matrix passes per fp32 product (weight gradient: always 6)
        for name, k1, k2, n in shapes:
            x1 = torch.randn(M, k1, device=dev)
            x2 = torch.randn(M, k2, device=dev) if k2 else None
            w1 = torch.randn(n, k1, device=dev) * 0.05
            w2 = torch.randn(n, k2, device=dev) * 0.05 if k2 else None
            b = torch.randn(n, device=dev)
            dy = torch.randn(M, n, device=dev)
            flop = 2.0 * M * n * (k1 + k2)
            for what, fn, fl in (("fwd", lambda: ops.linear_fwd(x1, w1, x2, w2, b, True), flop),
                                 ("bwd-data", lambda: ops.linear_bwd_data(dy, w1), 2.0 * M * n * k1),
                                 ("bwd-weight", lambda: ops.linear_bwd_weight(dy, x1, True), 2.0 * M * n * k1)):
                row = []
                for core in (1, 0):
                    was = ops.linear_set_bigtile(core)
                    ms = timeit(fn, args.reps)
                    ops.linear_set_bigtile(was)
                    row.append(ms)
                pc = 6 if (mode == 1 and what == "bwd-weight") else pieces
                print(f"mode {mode} {name:24s} {what:10s} core {row[0]:8.3f} ms = {fl / row[0] / 1e9:7.1f} TF fp32-eq "
                      f"({pc * fl / row[0] / 1e9:7.1f} executed)   tile {row[1]:8.3f} ms = {fl / row[1] / 1e9:7.1f} TF   "
                      f"x{row[1] / row[0]:.2f}", flush=True)
            del x1, x2, dy
        ops.linear_set_mode(prev)
    # the code assignment
    for (N, H, K, Dc) in ((M, 4, 512, 768), (43000, 4, 2048, 768)):
        xp = torch.randn(N, H * Dc, device=dev)
        embed = torch.nn.functional.normalize(torch.randn(H, K, Dc, device=dev), dim=-1).contiguous()
        esq = (embed * embed).sum(-1).contiguous()
        norm = torch.empty(N, H, device=dev)
        ind = torch.empty(N, H, dtype=torch.int64, device=dev)
        sq = torch.empty(1, device=dev)
        ws = torch.empty(int(lib.stemgnn_vq_workspace_bytes(N, H, Dc, K)), dtype=torch.uint8, device=dev)
        ops.linear_scratch(N, Dc, Dc, vq=(H, Dc, K))
        st = torch.cuda.current_stream().cuda_stream

        def run():
            check(lib.stemgnn_vq_assign_lean(xp.data_ptr(), N, H, Dc, embed.data_ptr(), esq.data_ptr(), K, norm.data_ptr(),
                                             ind.data_ptr(), sq.data_ptr(), 0.25, ws.data_ptr(), ws.numel(), st))
        fl = 2.0 * N * H * K * Dc
        row = []
        for core in (1, 0):
            was = ops.linear_set_bigtile(core)
            row.append(timeit(run, args.reps))
            ops.linear_set_bigtile(was)
        print(f"assign N={N} H={H} K={K} Dc={Dc}: core {row[0]:8.3f} ms = {fl / row[0] / 1e9:7.1f} TF fp32-eq "
              f"({(3 if ops.linear_set_pair(-1) else 6) * fl / row[0] / 1e9:7.1f} executed)   tile {row[1]:8.3f} ms = {fl / row[1] / 1e9:7.1f} TF   x{row[1] / row[0]:.2f}",
              flush=True)
        del xp


if __name__ == "__main__":
    main()
