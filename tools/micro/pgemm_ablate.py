# Round-2 record: ran against the library of commit b742edb, which still exported stemgnn_pgemm_* (tools/micro/pgemm.hip).
"""Ablations of k_pgemm_fwd (STEMGNN_PGEMM_DBG bits: 1 no MFMA, 2 no weight DMA, 4 no stores, 8 no activation DMA)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from pgemm_bench import timeit, dev
M = 102400
torch.manual_seed(0)
for (k1, n) in [(128, 128), (512, 128), (128, 512)]:
    a = torch.randn(M, k1, device=dev); w = torch.randn(n, k1, device=dev) * 0.1
    ap = ops.split_planes(a); wp = ops.weight_planes([w], [False])[0]
    t = timeit(lambda: ops.pgemm_fwd(ap, wp))
    print(f"dbg={os.environ.get('STEMGNN_PGEMM_DBG','0')} K={k1} N={n}: {t:7.1f} us")
