cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3tcp; rm -rf $out; mkdir -p $out
CASES=thin,dense REPS=2 timeout -k 10 200 python3 tools/perf_ranks.py > $out/plain.txt 2>&1; cat $out/plain.txt | tail -3
i=0
for ctrs in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  CASES=thin,dense REPS=2 timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -- python3 tools/perf_ranks.py > $out/p$i.txt 2>&1; echo "pass $i exit=$?"
  find $out/p$i -name "*kernel_trace.csv" -delete; find $out/p$i -name "*agent_info.csv" -delete
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r3tcp/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rank_loop_kernel" in row.get("Kernel_Name", ""):
            acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-32s n=%d values %s" % (c, len(v), ["%.4g" % x for x in v]))
PY
