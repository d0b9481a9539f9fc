#!/bin/bash
# rocprofv3 PMC passes (separate from kernel-trace; never combined with --sys-trace etc.)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc/p$i -- python3 bench.py --steps 100 --warmup 10 --profile-steps 0 --no-cpu-baseline --graph 0 > gpurun_out/pmc/p$i.json 2> gpurun_out/pmc/p$i.err
  echo "pass $i ($ctrs) exit=$?"
done
python3 tools/pmc_summary.py gpurun_out/pmc
