#!/bin/bash
# PMC passes over the frame-queue launch of rank_loop_kernel (tools/queue_bench.py: cfg2, 1e6 photons as 1025 lists, K frames in one launch):
#   tools/pmc_queue.sh [frames]      (through gpurun; the table goes to gpurun_out/pmcq/summary.txt)
# Counters in their own runs with --kernel-trace only; the program follows `--` directly.  Of the dispatches of rank_loop_kernel the one with the
# 512-workgroup grid is the K-frame queue launch (persistent workgroups, as many as the device holds; warm-up and timed call alike); the one-frame
# launches of the same run (1025 workgroups) are listed beside it.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
K=${1:-20}
export QUEUE_BENCH_WARM_FRAMES=$K     # (the queue kernel's two launches -- warm-up and timed -- are then the same work: persistent workgroups, one grid)
rm -rf gpurun_out/pmcq && mkdir -p gpurun_out/pmcq
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmcq/p$i -- python3 tools/queue_bench.py $K 1000000 1 > gpurun_out/pmcq/p$i.out 2> gpurun_out/pmcq/p$i.err
  echo "pass $i ($ctrs) exit=$?"
done
python3 - <<'PY' | tee gpurun_out/pmcq/summary.txt
import csv, glob
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))           # grid -> counter -> values
for f in glob.glob("gpurun_out/pmcq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rank_loop_kernel" not in row["Kernel_Name"]:
            continue
        rows[int(row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for grid in sorted(rows):
    print("rank_loop_kernel dispatches with Grid_Size %d (%d workgroups of 256 threads):" % (grid, grid // 256))
    for c, v in sorted(rows[grid].items()):
        print("    %-28s n=%3d mean=%.6g min=%.6g max=%.6g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
PY
find gpurun_out/pmcq -name "*kernel_trace.csv" -delete; find gpurun_out/pmcq -name "*agent_info.csv" -delete
find gpurun_out/pmcq -name "*counter_collection.csv" | while read f; do { head -1 "$f"; grep -E "rank_loop_kernel" "$f"; } > "$f.tmp"; mv "$f.tmp" "$f"; done
du -sh gpurun_out/pmcq
