"""K1 forward on C4 sampler batches (exact per-launch kernel time via the library's HIP-event stamps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd.data.synthetic import make_graph
from stem_gnn_amd.data.sampler import HipNeighborSampler
dev = torch.device("cuda:0")
g = make_graph(1_000_000, 20_000_000, 128, 4, kind="U", device=dev)
s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [10, 10], seed=5)
perm = torch.randperm(g.num_nodes, device=dev)
batches = [s.sample(perm[i * 1024:(i + 1) * 1024]) for i in range(20)]
xs = [g.node_text_feat[b.n_id] for b in batches]
augs = [b.graph.ensure_transpose().dropout_undirected(0.3) for b in batches]
for name, graphs in (("batch graph", [b.graph for b in batches]), ("augmented graph", augs)):
    for rep in range(2):
        ops.k1_timer.reset(True)
        for _ in range(10):
            for gr, x in zip(graphs, xs):
                ops.sage_agg_fwd(x, gr, None, g.edge_text_feat)
        torch.cuda.synchronize()
        ms, n, by = ops.k1_timer.collect()
        ops.k1_timer.reset(False)
    print(f"R={os.environ.get('STEMGNN_K1_R', '1')} {name}: {n} launches, avg {ms / n * 1e3:.2f} us, {by / ms / 1e6:.0f} GB/s "
          f"({by / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
