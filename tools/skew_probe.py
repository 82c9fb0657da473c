"""K1/K2 on a uniform vs a Zipf-skewed graph (C2 size), with and without heavy-row splitting."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd.data.synthetic import make_graph
from stem_gnn_amd.graph import GraphStructure
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for kind in ("U", "Z"):
    g = make_graph(100_000, 1_000_000, 128, 4, kind=kind, device=dev)
    x = torch.randn(100_000, 128, device=dev)
    ga = torch.randn(100_000, 128, device=dev)
    by = ops.k1_algorithmic_bytes(100_000, 1_000_000, 128, "table", 4)
    for label in ("one pass", "split"):
        gs = GraphStructure(g.edge_index, 100_000, g.xe).ensure_transpose()
        d_in, d_out = gs.max_in_degree, gs.max_out_degree
        if label == "one pass":
            gs.max_in_degree = gs.max_out_degree = 0
        else:
            gs.max_in_degree = gs.max_out_degree = None
        us = timeit(lambda: ops.sage_agg_fwd(x, gs, None, g.edge_text_feat))
        usb = timeit(lambda: ops.sage_agg_bwd(ga, x, gs, None, g.edge_text_feat))
        extra = ""
        if gs._plan_in is not None:
            extra = f" items/heavy in {gs._plan_in.counts.tolist()} out {gs._plan_out.counts.tolist()}"
        print(f"Graph-{kind} [{label}]: max degree in {d_in} out {d_out}, K1 fwd {us:.1f} us ({by/us/1e3:.0f} GB/s), "
              f"K2 bwd {usb:.1f} us{extra}", flush=True)
