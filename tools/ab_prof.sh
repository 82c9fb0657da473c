#!/bin/bash
# Same-box per-kernel A/B: rocprofv3 kernel stats of the working tree and of the HEAD checkout under _ab/.
set -e
R=$(pwd)
mkdir -p $R/gpurun_out/abp
cd /tmp && export TMPDIR=/tmp
for w in tree head; do
  if [ $w = head ]; then D=$R/_ab; else D=$R; fi
  cd $D
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$w -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --e2e-steps 0 --pmc-traffic off > /tmp/abp_$w.log 2>&1
  cp $(find /tmp/abp_$w -name '*kernel_stats.csv' | head -1) $R/gpurun_out/abp/${w}_kernel_stats.csv
done
cd $R
python - <<'PY'
import csv
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        d[r['Name'][:70]]=(int(r['Calls']),float(r['TotalDurationNs']))
    return d
a=load('gpurun_out/abp/head_kernel_stats.csv'); b=load('gpurun_out/abp/tree_kernel_stats.csv')
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0)); cb,tb=b.get(k,(0,0))
    rows.append((tb-ta,k,ca,ta,cb,tb))
rows.sort(reverse=True)
print('delta_us_per_step(25 steps) name calls_head tot_head_us calls_tree tot_tree_us')
for d,k,ca,ta,cb,tb in rows[:14]+rows[-6:]:
    print(f'{d/25e3:8.2f} {k:70s} {ca:6d} {ta/1e3:10.1f} {cb:6d} {tb/1e3:10.1f}')
print('total', sum(v[1] for v in a.values())/25e6, sum(v[1] for v in b.values())/25e6)
PY
