#!/bin/bash
# rocprofv3 kernel trace of a short bench run; summaries are copied to profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps "${1:-500}" --warmup 20 --profile-steps 0 --no-cpu-baseline --graph "${2:-0}" > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
echo "exit=$?"
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -20
cat gpurun_out/prof/bench.json
