#!/bin/bash
# instruction-cache counters of rank_loop_kernel on the benchmark frame (separate --pmc passes, --kernel-trace only; the program follows `--`)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/icache
rm -rf $out && mkdir -p $out
i=0
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "InstrFetchLatency" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  echo "== pass $i ($ctrs)"
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --steps 5 --warmup 1 --other-mode 0 --host-driver 0 --no-cpu-baseline --shared-clock-rounds 0 $EXTRA > $out/p$i.json 2> $out/p$i.err; echo "exit=$?"
done
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/icache/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rank_loop" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-30s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
