#!/bin/bash
# rocprofv3 kernel trace of a short bench run + per-step breakdown.  usage: tools/prof_step.sh <tag> [bench args...]
# (run on the GPU box through gpurun; writes gpurun_out/<tag>_step.txt, _seq.txt and <tag>_kernel_stats.csv)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --e2e-steps 0 --pmc-traffic off "$@" > "gpurun_out/${tag}_prof.log" 2>&1
echo "rocprof rc=$?"
trace=$(find "$out" -name "*kernel_trace.csv" | head -1)
stats=$(find "$out" -name "*kernel_stats.csv" | head -1)
echo "trace=$trace stats=$stats"
python tools/step_breakdown.py "$trace" 70 > "gpurun_out/${tag}_step.txt"
python tools/step_sequence.py "$trace" > "gpurun_out/${tag}_seq.txt" 2>/dev/null
cp "$stats" "gpurun_out/${tag}_kernel_stats.csv"
rm -rf "$out"
cat "gpurun_out/${tag}_step.txt"
