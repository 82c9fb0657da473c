"""Per-step GPU time trend over many steps (same C4-like batch re-used, or distinct batches)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops
from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure, set_validation
from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step
dev = torch.device("cuda:0")
D = 128
params = default_params(); params.update(input_dim=D, hidden_dim=D, code_dim=D)
torch.manual_seed(0)
N, E = 102400, 112000
x = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1)
ei = torch.stack([torch.randint(0, N, (E,), device=dev), torch.randint(0, 11264, (E,), device=dev)])
xe = torch.randint(0, 4, (E,), device=dev)
tab = torch.nn.functional.normalize(torch.randn(4, D, device=dev), dim=-1)
g = GraphStructure(ei, N, xe, validate=False).ensure_transpose()
model = build_model(params, dev); opt, sched = build_optimizer(model, params)
set_validation(False); model.train()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
t0 = time.perf_counter()
evs[0].record()
host = []
for i in range(n):
    h0 = time.perf_counter()
    pretrain_step(model, opt, sched, params, x, g, EdgeTypeAttr(tab, xe), 1024, record_draws=False)
    host.append(time.perf_counter() - h0)
    evs[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for a in range(0, n, 20):
    print(f"steps {a:3d}-{a+19:3d}: gpu-span {sum(gpu[a:a+20])/20:.3f} ms  host-issue {sum(host[a:a+20])/20*1e3:.3f} ms")
print(f"wall per step {wall / n * 1e3:.3f} ms; mem allocated {torch.cuda.memory_allocated()/1e9:.2f} GB reserved {torch.cuda.memory_reserved()/1e9:.2f} GB")
