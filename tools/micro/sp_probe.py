# Round-3 record: ran against a library build that still contained tools/micro/spgemm.hip (stemgnn_linear_set_sp).
"""Specialised-wave products (csrc/spgemm.hip) beside the tile / weight-stationary kernels on C4-batch shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd._lib import lib
from kbench import timeit, dev
M = 102400
torch.manual_seed(0)
for (k1, k2, n, what) in [(128, 128, 128, "layer product + stats"), (128, 0, 512, "project_in"), (128, 0, 128, "N=K=128")]:
    a = torch.randn(M, k1, device=dev); w = torch.randn(n, k1, device=dev) * 0.1
    a2 = torch.randn(M, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.1 if k2 else None
    b = torch.randn(n, device=dev)
    res = {}
    for name, (ws, sp) in {"tile": (0, 0), "ws": (128, 0), "sp": (0, 1), "default": (128, 256)}.items():
        lib.stemgnn_linear_set_ws(ws); lib.stemgnn_linear_set_sp(sp)
        res[name] = timeit(lambda: ops.linear_fwd(a, w, a2, w2, b, k2 > 0), iters=30)
    print(f"fwd {what} M={M} K={k1}+{k2} N={n}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
for (n, k, what) in [(512, 128, "project_in bwd-data"), (128, 128, "layer bwd-data"), (128, 512, "project_out-like bwd-data")]:
    dy = torch.randn(M, n, device=dev); w = torch.randn(n, k, device=dev) * 0.1
    res = {}
    for name, (ws, sp) in {"tile": (0, 0), "ws": (128, 0), "sp": (0, 1), "default": (128, 256)}.items():
        lib.stemgnn_linear_set_ws(ws); lib.stemgnn_linear_set_sp(sp)
        res[name] = timeit(lambda: ops.linear_bwd_data(dy, w), iters=30)
    print(f"bwd-data {what} M={M} N={n} -> K={k}: " + "  ".join(f"{k_} {v:6.1f} us" for k_, v in res.items()), flush=True)
