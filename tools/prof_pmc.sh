#!/bin/bash
# Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + one kernel trace of the same short bench run, reduced to the
# per-step traffic table.  usage: tools/prof_pmc.sh <tag>   (GPU box; writes gpurun_out/<tag>_step_traffic.{csv,json})
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 10 --warmup 3 --no-cpu-baseline --e2e-steps 0 --no-extra --pmc-traffic off"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o $c -- python3 bench.py $args > gpurun_out/${tag}_pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
rm -rf gpurun_out/prof_${tag}t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${tag}t -o t -- python3 bench.py $args > gpurun_out/${tag}_trace.log 2>&1
trace=$(find gpurun_out/prof_${tag}t -name "*kernel_trace.csv" | head -1)
python tools/step_traffic.py "$trace" gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/${tag}_step_traffic.csv gpurun_out/${tag}_step_traffic.json
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/prof_${tag}t
