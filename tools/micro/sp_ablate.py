# Round-3 record: ran against a library build that still contained tools/micro/spgemm.hip (stemgnn_linear_set_sp).
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stem_gnn_amd import ops
from stem_gnn_amd._lib import lib
from kbench import timeit, dev
M = 102400
torch.manual_seed(0)
lib.stemgnn_linear_set_ws(0); lib.stemgnn_linear_set_sp(1)
for (k1, k2, n, what) in [(128, 128, 128, "layer"), (128, 0, 512, "project_in"), (128, 0, 128, "N=K=128")]:
    a = torch.randn(M, k1, device=dev); w = torch.randn(n, k1, device=dev) * 0.1
    a2 = torch.randn(M, k2, device=dev) if k2 else None
    w2 = torch.randn(n, k2, device=dev) * 0.1 if k2 else None
    b = torch.randn(n, device=dev)
    t = timeit(lambda: ops.linear_fwd(a, w, a2, w2, b, False), iters=30)
    print(f"dbg={os.environ.get('STEMGNN_SP_DBG','0')} {what}: {t:6.1f} us", flush=True)
