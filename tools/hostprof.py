"""cProfile of the host side of the pretraining step (C4 batch shapes)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops  # noqa: E402
from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure, set_validation  # noqa: E402
from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step  # noqa: E402

dev = torch.device("cuda:0")
D = 128
params = default_params()
params.update(input_dim=D, hidden_dim=D, code_dim=D)
torch.manual_seed(0)
N, E = 102400, 112000
x = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1)
ei = torch.stack([torch.randint(0, N, (E,), device=dev), torch.randint(0, 11264, (E,), device=dev)])
xe = torch.randint(0, 4, (E,), device=dev)
tab = torch.nn.functional.normalize(torch.randn(4, D, device=dev), dim=-1)
g = GraphStructure(ei, N, xe, validate=False).ensure_transpose()
model = build_model(params, dev)
opt, sched = build_optimizer(model, params)
set_validation(False)
model.train()


def step():
    pretrain_step(model, opt, sched, params, x, g, EdgeTypeAttr(tab, xe), 1024, record_draws=False)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / 20 * 1e3:.3f} ms; with final sync {(t2 - t0) / 20 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
