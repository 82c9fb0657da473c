"""How fast is the library's bf16 GEMM on the six-piece-product shape (K' = 6 K, fp32 accumulate) against this
repository's exact-piece tile kernel at the D = 768 shapes?  (torch.mm -> hipBLASLt / rocBLAS.)"""
import os, sys, time
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


for M, K, N in ((169343, 768, 3072), (169343, 768, 768), (169343, 3072, 768), (102400, 128, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    mine = timed(lambda: ops.linear_fwd(x, w, None, None, None))
    xa = torch.randn(M, 6 * K, device=dev, dtype=torch.bfloat16)
    wb = torch.randn(N, 6 * K, device=dev, dtype=torch.bfloat16)
    lib = timed(lambda: torch.mm(xa, wb.t()))  # bf16 out; the fp32-out variant below
    try:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
        lib32 = timed(lambda: torch.mm(xa, wb.t(), out_dtype=torch.float32))
    except Exception as e:  # noqa: BLE001
        lib32 = float("nan")
    x3 = torch.randn(M, K, device=dev)
    f32 = timed(lambda: torch.mm(x3, w.t()))
    tf = 2.0 * M * K * N / 1e12
    print(f"M={M} K={K} N={N}: tile kernel {mine:7.3f} ms ({tf / mine * 1e3:6.1f} TF/s fp32-eq) | library bf16 K'=6K {lib:7.3f} ms "
          f"({6 * tf / lib * 1e3:7.1f} TF/s bf16), fp32 out {lib32:7.3f} ms | library fp32 {f32:7.3f} ms ({tf / f32 * 1e3:6.1f} TF/s)")
    del x, w, xa, wb, x3
