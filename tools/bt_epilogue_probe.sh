# Store-cost probe of the big-tile core's forward epilogue (measurement switches of csrc/bigtile.hip, STEMGNN_BT_DBG: 1 no
# output stores, 2 no statistics, 4 every second store instruction skipped, 16 non-temporal stores), on the three-bf16-piece
# form (STEMGNN pair format off: the six-pass product the probe was designed on), with (169343 rows: 7944 tiles on 256 CUs)
# and without (163840 rows: 7680 = 30 x 256 tiles) a last partial round of tiles.
#   usage (GPU box): bash tools/bt_epilogue_probe.sh > gpurun_out/bt_epilogue_probe.log
for dbg in 0 1 2 3 6 18; do for rows in 169343 163840; do echo "dbg=$dbg rows=$rows"; STEMGNN_BT_DBG=$dbg python tools/bt_bench.py --modes 1 --rows $rows --reps 5 --pair 0 2>&1 | grep "project_in.*fwd \|lin 768->768 *fwd" | cut -c1-120; done; done
