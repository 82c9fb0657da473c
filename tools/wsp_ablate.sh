#!/bin/bash
# Phase ablation of the pair weight-stationary kernel (csrc/wspair.hip, backward-data instance): STEMGNN_WSP_DBG bits
# 1 = no matrix instructions, 2 = no cut, 4 = no epilogue (no stores), 8 = no prefetch loads (32 alone = the instrumented instance, nothing removed).  Results are wrong by design.
for d in ${DBGS:-0 1 2 4 8 3 5 6 7 15}; do
  STEMGNN_WSP_DBG=$d python3 - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from stem_gnn_amd import ops
dev = torch.device("cuda:0")
out = []
for M in (32768, 101927, 262144):
    dy = torch.randn(M, 128, device=dev); w = torch.randn(128, 128, device=dev)
    for _ in range(3): ops.linear_bwd_data(dy, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear_bwd_data(dy, w)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
print("dbg", os.environ.get("STEMGNN_WSP_DBG"), "  ".join(out), flush=True)
PY
done
