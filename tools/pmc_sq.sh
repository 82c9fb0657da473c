#!/bin/bash
# SQ stall / activity counters per kernel for tools/kbench.py linear (two rocprofv3 --pmc passes, nothing else traced).
# usage (GPU box): tools/pmc_sq.sh   -> gpurun_out/pmc_sq.txt
R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmcsq_$i
  rocprofv3 --pmc $P --output-format csv -d /tmp/pmcsq_$i -o p -- python3 tools/kbench.py linear > /tmp/pmcsq_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY' > gpurun_out/pmc_sq.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for i in (1,2):
    for f in glob.glob(f'/tmp/pmcsq_{i}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES','SQ_ACTIVE_INST_VALU'): cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'linear' not in k: continue
    wc=v.get('SQ_WAVE_CYCLES',1)
    print(k)
    for c,val in sorted(v.items()):
        print(f'   {c:32s} {val:14.3e}  {val/wc:7.3f} of wave cycles')
PY
cat gpurun_out/pmc_sq.txt
