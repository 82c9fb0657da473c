#!/bin/bash
# Same-box A/B: the working tree against a checkout of HEAD built under _ab/ (git-ignored).
set -e
mkdir -p gpurun_out
for i in 1 2 3; do
  (cd _ab && python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --e2e-steps 0 2>/dev/null) > gpurun_out/ab_head_$i.json
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --e2e-steps 0 --pmc-traffic off 2>/dev/null > gpurun_out/ab_tree_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
