"""K1 forward on C4 sampler batches over the rows that can receive edges (what the encoder phase launches), exact
per-launch kernel time via the library's HIP-event stamps.  STEMGNN_K1_PROBE selects the launch shape."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd._lib import lib, check
from stem_gnn_amd.data.synthetic import make_graph
from stem_gnn_amd.data.sampler import HipNeighborSampler
dev = torch.device("cuda:0")
g = make_graph(1_000_000, 20_000_000, 128, 4, kind="U", device=dev)
s = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [10, 10], seed=5)
perm = torch.randperm(g.num_nodes, device=dev)
batches = [s.sample(perm[i * 1024:(i + 1) * 1024]) for i in range(20)]
xs = [g.node_text_feat[b.n_id] for b in batches]
augs = [b.graph.ensure_transpose().dropout_undirected(0.2) for b in batches]
junk = torch.empty(64 * 1024 * 1024, device=dev)  # 256 MB: flushes the Infinity Cache between launches when asked
for name, graphs in (("batch graph", [b.graph for b in batches]), ("augmented graph", augs)):
    for cold in (False, True):
        for rep in range(2):
            ops.k1_timer.reset(True)
            for _ in range(5):
                for gr, x in zip(graphs, xs):
                    A = gr.active_rows
                    agg = torch.empty(A, 128, device=dev)
                    if cold:
                        junk.add_(1.0)
                    check(lib.stemgnn_sage_agg_fwd(x.data_ptr(), A, 128, gr.rowptr.data_ptr(), gr.src.data_ptr(),
                                                   gr.eid.data_ptr(), None, g.edge_text_feat.data_ptr(),
                                                   gr.etype_slot.data_ptr(), 4, agg.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream))
                    ops.k1_timer.bytes.append((A, gr, 128, "table", 4))
            torch.cuda.synchronize()
            ms, n, by = ops.k1_timer.collect()
            ops.k1_timer.reset(False)
        print(f"variant={os.environ.get('STEMGNN_K1_PROBE', '0')} {name} {'cold' if cold else 'warm'}: {n} launches, avg {ms / n * 1e3:.2f} us, "
              f"{by / n / 1e6:.1f} MB, {by / ms / 1e6:.0f} GB/s ({by / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
