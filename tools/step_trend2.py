"""Per-step time trend with DISTINCT sampled batches (as bench.py does)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader
from stem_gnn_amd.data.synthetic import make_graph
from stem_gnn_amd.graph import EdgeTypeAttr, set_validation
from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step
dev = torch.device("cuda:0")
D = 128
params = default_params(); params.update(input_dim=D, hidden_dim=D, code_dim=D)
g = make_graph(1_000_000, 20_000_000, D, 4, kind="U", device=dev)
sampler = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [10, 10], seed=1)
loader = iter(NeighborLoader(sampler, torch.arange(g.num_nodes, device=dev), 1024, shuffle=True, seed=7))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
batches = []
for _ in range(n):
    b = next(loader)
    batches.append((ops.gather_rows(g.node_text_feat, b.x.contiguous()), b.graph.ensure_transpose(), b.xe, b.batch_size))
model = build_model(params, dev); opt, sched = build_optimizer(model, params)
set_validation(False); model.train()
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
host = []
for i in range(n):
    x, gs, xe, bs = batches[i]
    h0 = time.perf_counter()
    pretrain_step(model, opt, sched, params, x, gs, EdgeTypeAttr(g.edge_text_feat, xe), bs, record_draws=False)
    host.append(time.perf_counter() - h0)
    evs[i + 1].record()
torch.cuda.synchronize()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for a in range(0, n, 10):
    print(f"steps {a:3d}-{a+9:3d}: gpu-span {sum(gpu[a:a+10])/10:.3f} ms  host-issue {sum(host[a:a+10])/10*1e3:.3f} ms  N {batches[a][0].size(0)}")
print(f"mem allocated {torch.cuda.memory_allocated()/1e9:.2f} GB reserved {torch.cuda.memory_reserved()/1e9:.2f} GB")
print({k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "num_ooms")})
