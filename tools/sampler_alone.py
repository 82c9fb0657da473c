"""The sampler alone on the C4 graph (no step beside it): wall per batch and, under `rocprofv3 --kernel-trace --stats`,
its kernels' own durations.  usage: python tools/sampler_alone.py [batches]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader
from stem_gnn_amd.data.synthetic import make_graph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
g = make_graph(1_000_000, 20_000_000, 128, 4, kind="U", device=dev, graph_seed=1234, feat_seed=0)
s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [10, 10], seed=100)
loader = NeighborLoader(s, torch.arange(g.num_nodes, device=dev), 1024, shuffle=True, seed=7)
it = iter(loader)
for _ in range(5):
    b = next(it); ops.gather_rows(g.node_text_feat, b.x, validate=False, capacity=b.cap_nodes)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    b = next(it); ops.gather_rows(g.node_text_feat, b.x, validate=False, capacity=b.cap_nodes)
torch.cuda.synchronize()
print(f"sampler + feature gather, one batch at a time: {(time.perf_counter() - t) / n * 1e3:.3f} ms per batch")
