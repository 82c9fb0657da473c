#!/bin/bash
# VALU : MFMA instruction ratio of the product kernels over a short C4 bench run (one rocprofv3 --pmc pass).
R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf /tmp/pmcvm
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d /tmp/pmcvm -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --e2e-steps 0 --pmc-traffic off --preheat off --no-dense-profile $BENCH_ARGS > /tmp/pmcvm.log 2>&1
echo "rc=$?"
python3 - <<'PY' > gpurun_out/pmc_valu_mfma${TAG}.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmcvm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('stemgnn::(anonymous namespace)::','').replace('void ','')[:52]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
print('kernel | VALU(incl. MFMA) MFMA  (VALU-MFMA)/MFMA | LDS/MFMA  VMEM/MFMA')
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_INSTS_MFMA',0)):
    m=v.get('SQ_INSTS_MFMA',0)
    if m<=0: continue
    va=v.get('SQ_INSTS_VALU',0)
    print(f"{k:52s} | {va:12.3e} {m:12.3e} {(va-m)/m:6.2f} | {v.get('SQ_INSTS_LDS',0)/m:5.2f} {(v.get('SQ_INSTS_VMEM_RD',0)+v.get('SQ_INSTS_VMEM_WR',0))/m:5.2f}")
PY
cat gpurun_out/pmc_valu_mfma${TAG}.txt
