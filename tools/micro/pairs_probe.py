"""fp16-pair forward products against the exact three-piece form: error against fp64 and launch time.
usage (GPU box): python tools/micro/pairs_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stem_gnn_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, K, N, scale in (("project_in", 102400, 128, 512, 1.0), ("head", 102400, 128, 128, 1.0),
                             ("tiny values", 16384, 128, 128, 1e-5), ("wide range", 16384, 128, 128, None)):
    x = torch.randn(M, K, device=dev)
    if scale is None:
        x = x * torch.exp(torch.randn(M, K, device=dev) * 4)  # magnitudes over ~10 decades
    else:
        x = x * scale
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = (x.double() @ w.double().t() + b.double())
    den = (x.double().abs() @ w.double().abs().t() + b.double().abs())  # what a relative rounding step acts on
    for pairs in (0, 1):
        ops.linear_set_pairs(pairs)
        y = ops.linear_fwd(x, w, None, None, b)[0]
        err = ((y.double() - ref).abs() / den).max().item()
        us = timed(lambda: ops.linear_fwd(x, w, None, None, b))
        print(f"{name:12s} M={M} K={K} N={N} pairs={pairs}: max |err| / sum|x||w| = {err:.3e}   {us:7.1f} us")
ops.linear_set_pairs(0)
