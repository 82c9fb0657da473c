import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for K in (128, 256, 512):
    for M in (65536, 98304, 102400, 131072, 196608):
        a = torch.randn(M, K, device=dev); w = torch.randn(128, K, device=dev); b = torch.randn(128, device=dev)
        us = timeit(lambda: ops.linear_fwd(a, w, None, None, b, False))
        print(f"K={K} M={M} tiles={M//128}: {us:7.1f} us  {2.0*M*K*128/us/1e6:6.1f} TF/s")
