#!/bin/bash
# rocprofv3 kernel stats of a short C4 bench run, top kernels per step.  usage: tools/prof_top.sh [ENV=VAL ...]
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $R
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --e2e-steps 0 > /tmp/pt.log 2>&1
cp $(find /tmp/pt -name '*kernel_stats.csv' | head -1) $R/gpurun_out/top_kernel_stats.csv
python - <<'PY'
import csv
rows=[]
for r in csv.DictReader(open('gpurun_out/top_kernel_stats.csv')):
    rows.append((float(r['TotalDurationNs'])/25e3, int(r['Calls'])/25, float(r['AverageNs'])/1e3, r['Name'][:100]))
rows.sort(reverse=True)
for t,c,a,n in rows[:28]: print(f'{t:8.1f} {c:5.1f} {a:7.1f} {n}')
print('total', sum(r[0] for r in rows))
PY
