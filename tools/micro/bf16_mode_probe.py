"""bf16 GEMM mode (stemgnn_linear_set_mode(2): operands rounded to bf16 while staged, one matrix pass) against the vendor
library's bf16 GEMM on pre-rounded operands (+ the rounding pass it would need), at the D = 768 shapes of C5."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stem_gnn_amd import ops

dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


prev = ops.linear_set_mode(2)
for M, K, N in ((108000, 768, 3072), (108000, 768, 768), (108000, 3072, 768), (108000, 1536, 768)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    mine = timed(lambda: ops.linear_fwd(x, w, None, None, None))
    xb, wb = x.bfloat16(), w.bfloat16()
    lib = timed(lambda: torch.mm(xb, wb.t(), out_dtype=torch.float32))
    cvt = timed(lambda: x.bfloat16())
    tf = 2.0 * M * K * N / 1e12
    print(f"M={M} K={K} N={N}: mode-2 tile kernel {mine:6.3f} ms ({tf / mine * 1e3:6.0f} TF/s) | library bf16, fp32 out {lib:6.3f} ms "
          f"({tf / lib * 1e3:6.0f} TF/s) + rounding pass of x {cvt:6.3f} ms")
ops.linear_set_mode(prev)
