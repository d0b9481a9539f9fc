#!/bin/bash
# Instruction-mix and wait counters of rank_loop_kernel on the benchmark frame (separate --pmc passes, --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_valu
rm -rf $out && mkdir -p $out
i=0
for ctrs in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  echo "== pass $i: $ctrs"
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --pools 1 --host-driver 0 --steps 5 --warmup 1 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/p$i.json 2> $out/p$i.err; echo "exit=$?"
done
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_valu/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rank_loop_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s launches %3d  mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
