# store-cost probe of the big-tile core's forward epilogue (STEMGNN_BT_DBG: 1 no stores, 2 no statistics, 4 every second
# store instruction skipped, 16 non-temporal stores)
for dbg in 2 3 6 18; do echo "dbg=$dbg"; STEMGNN_BT_DBG=$dbg python tools/bt_bench.py --modes 1 --rows 163840 --reps 5 2>&1 | grep "project_in.*fwd " | cut -c1-120; done
