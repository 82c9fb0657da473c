"""Print per-step, per-term relative differences between the HIP path and the CPU oracle (a test aid: it lives
under tests/ because only tests may use the oracle).  Run on the GPU box: python tests/parity_probe.py"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import stem_oracle as O
from test_gpu_model import make_models
from stem_gnn_amd import ops
from stem_gnn_amd.graph import EdgeTypeAttr
from stem_gnn_amd.pretrain import pretrain_step, default_params
from stem_gnn_amd.utils.others import get_scheduler
dev = torch.device("cuda:0")
N, E, D, L, H, K = 600, 5000, 64, 2, 4, 64
bs = 200
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
om, gm = make_models(D, L, H, K, D, dev)
params = default_params(); params.update(pretrain_lr=lr)
torch.manual_seed(11)
x = torch.nn.functional.normalize(torch.randn(N, D), dim=-1)
half = torch.randint(0, N, (2, E // 2))
ei = torch.cat([half, half.flip(0)], dim=1)[:, torch.randperm(E)]
table = torch.nn.functional.normalize(torch.randn(4, D), dim=-1)
et = torch.randint(0, 4, (E,))
opt_o = torch.optim.AdamW(om.parameters(), lr=lr, weight_decay=1e-5)
opt_g = torch.optim.AdamW(gm.parameters(), lr=lr, weight_decay=1e-5)
sch_o, sch_g = get_scheduler(opt_o, True, 50), get_scheduler(opt_g, True, 50)
ops.manual_seed(99)
for step in range(8):
    loss_g, losses_g, draws = pretrain_step(gm, opt_g, sch_g, params, x.to(dev), ei.to(dev), EdgeTypeAttr(table.to(dev), et.to(dev)), bs)
    cpu_draws = {k: ([m.cpu() for m in v] if isinstance(v, list) else v.cpu()) for k, v in draws.items()}
    loss_o, losses_o, ind_o = O.pretrain_step(om, opt_o, sch_o, params, x, ei, table[et], bs, cpu_draws)
    rel = {k: float((losses_g[k].cpu().reshape(-1) - losses_o[k].reshape(-1)).abs() / losses_o[k].reshape(-1).abs().clamp(min=1e-12)) for k in losses_o if k != "env_reg_loss"}
    print(step, "total %.6f rel %.2e" % (float(loss_o), float((loss_g.cpu().reshape(-1) - loss_o.reshape(-1)).abs() / loss_o.abs())), {k: "%.1e" % v for k, v in rel.items()})
    pd = max(float((p2.detach().cpu() - p1.detach()).abs().max()) for (n1, p1), (n2, p2) in zip(om.named_parameters(), gm.named_parameters()) if not n1.endswith("lin_l.bias"))
    print("   max param diff (excl lin_l.bias): %.2e" % pd)
