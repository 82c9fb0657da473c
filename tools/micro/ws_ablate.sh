#!/bin/bash
# Round-2 record: the STEMGNN_WS_DBG ablation bits of k_linear_ws existed in the library of commit b742edb only.
# Ablation of the weight-stationary product (csrc/wsgemm.hip): STEMGNN_WS_DBG bits 1 = no activation loads, 2 = no cut /
# LDS writes, 4 = no matrix instructions, 8 = no output stores.  Results are wrong by construction; times only.
for d in 0 1 2 4 8 3 12 15 7 11; do
  echo "dbg=$d $(STEMGNN_WS_DBG=$d python tools/kbench.py linear 2>&1 | grep 'hip' | head -1)"
done
