#!/bin/bash
# PMC passes for rank_loop_kernel (ranks mode of bench.py)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcr && mkdir -p gpurun_out/pmcr
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmcr/p$i -- python3 bench.py --steps 5 --warmup 1 --other-mode 0 --no-cpu-baseline > gpurun_out/pmcr/p$i.json 2> gpurun_out/pmcr/p$i.err
  echo "pass $i ($ctrs) exit=$?"
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmcr/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void mcrat::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "rank_loop" in k:
        print(k)
        for c, v in sorted(acc[k].items()):
            print("    %-24s n=%3d mean=%.6g min=%.6g max=%.6g" % (c, len(v), sum(v)/len(v), min(v), max(v)))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmcr/kt -- python3 bench.py --steps 20 --warmup 2 --other-mode 0 --no-cpu-baseline > gpurun_out/pmcr/kt.json 2> gpurun_out/pmcr/kt.err
find gpurun_out/pmcr/kt -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-220
