#!/bin/bash
# SQ stall / activity counters per kernel over a short C4 bench run (two rocprofv3 --pmc passes, nothing else traced).
# usage (GPU box): tools/pmc_sq_step.sh [name-substring ...]   -> gpurun_out/pmc_sq_step${TAG}.txt
# PMC_PROG="tools/wsp_bench.py --reps 3": another program than the bench
R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmcsq_$i
  rocprofv3 --pmc $P --output-format csv -d /tmp/pmcsq_$i -o p -- python3 ${PMC_PROG:-bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --e2e-steps 0 --pmc-traffic off --preheat off --no-dense-profile $BENCH_ARGS} > /tmp/pmcsq_$i.log 2>&1
  echo "pass $i rc=$?"
done
FILTER="$*" python3 - <<'PY' > gpurun_out/pmc_sq_step${TAG}.txt
import csv,glob,collections,os
flt=os.environ.get('FILTER','').split()
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for i in (1,2):
    for f in glob.glob(f'/tmp/pmcsq_{i}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].replace('stemgnn::(anonymous namespace)::','').replace('void ','')[:48]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
rows=sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))
print('kernel launches | per wave-cycle: wait_any wait_inst wait_lds act_valu act_lds act_vmem | mfma_busy/(wave_cycles*4/2) coexec/mfma_busy | lds_conflict/wave_cycle')
for k,v in rows[:40]:
    if flt and not any(s in k for s in flt): continue
    wc=v.get('SQ_WAVE_CYCLES',1) or 1
    mb=v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)
    print(f"{k:48s} {n[k]:4d} | {v.get('SQ_WAIT_ANY',0)/wc:5.2f} {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} {v.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} {v.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} {v.get('SQ_ACTIVE_INST_VMEM',0)/wc:5.2f} | {mb/(wc*2):5.2f} {(v.get('SQ_VALU_MFMA_COEXEC_CYCLES',0)/mb if mb else 0):5.2f} | {v.get('SQ_LDS_BANK_CONFLICT',0)/wc:5.2f}")
PY
cat gpurun_out/pmc_sq_step${TAG}.txt
