#!/bin/bash
# Kernel trace of the step with the loader inside the loop -> per-batch loader breakdown (tools/loader_breakdown.py).
# usage (on the GPU box): bash tools/loader_prof.sh [E2E_STEPS]
n=${1:-20}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d /tmp/lp -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --e2e-steps $n --pmc-traffic off > gpurun_out/loader_prof_bench.log 2>&1 || exit 1
f=$(find /tmp/lp -name '*kernel_trace.csv' | head -1)
python tools/loader_breakdown.py "$f" $n > gpurun_out/loader_breakdown.txt
cat gpurun_out/loader_breakdown.txt
