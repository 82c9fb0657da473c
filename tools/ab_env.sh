#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_env.sh VAR=VALUE  (bench.py C4, two runs each way)
set -e
mkdir -p gpurun_out
for i in 1 2; do
  env "$1" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --e2e-steps 0 2>/dev/null > gpurun_out/abe_with_$i.json
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --e2e-steps 0 2>/dev/null > gpurun_out/abe_without_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abe_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'])
PY
