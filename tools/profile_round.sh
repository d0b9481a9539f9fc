#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  tools/profile_round.sh  (run through gpurun), then
# `python tools/pmc_to_json.py gpurun_out/round profiles r04` here to turn the merged output into the committed files.
# PMC passes are separate runs with --kernel-trace only (never with --sys-trace & co); the program follows `--` directly.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round
part=${PART:-ab}        # PART=a: the headline's bench, traces and counters; PART=b: cfg3, cfg5, producers and consumers (two GPU calls of ~10 minutes)
mkdir -p $out
if [[ $part == *a* ]]; then
echo "== bench, no profiler"
python3 bench.py > $out/bench.json 2> $out/bench.err; echo "exit=$?"
echo "== kernel trace of the default bench command (frame queue: the timed region is ONE launch of rank_loop_kernel; the per-dispatch trace of that kernel is kept)"
# (--warmup 20: the warm-up is then a 20-frame launch like the timed one, so the --stats average of the queue kernel IS the duration of such a launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_default -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0 > $out/kt_default.json 2> $out/kt_default.err; echo "exit=$?"
python3 tools/queue_dispatches.py $out/kt_default rank_loop_kernel $out/queue_dispatches.csv
find $out/kt_default -name "*kernel_trace.csv" -delete
echo "== kernel trace, one launch per frame on one pool (round 3's kernel-alone figure)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_oneframe -- python3 bench.py --launch-shape pools --pools 1 --no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0 > $out/kt_oneframe.json 2> $out/kt_oneframe.err; echo "exit=$?"
echo "== kernel trace of the HEADLINE's launch shape: three pools on three streams (the per-dispatch trace is kept until the union of the overlapping launches is taken)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_pools3 -- python3 bench.py --launch-shape pools --pools 3 --steps 20 --warmup 5 --no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0 > $out/kt_pools3.json 2> $out/kt_pools3.err; echo "exit=$?"
python3 tools/union_busy.py $out/kt_pools3 rank_loop_kernel 20 5 3 $out/pools3_union.json
find $out/kt_pools3 -name "*kernel_trace.csv" -delete
echo "== kernel trace, list mode (eager launches so that every kernel is a trace record)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_list -- python3 bench.py --mode list --steps 500 --warmup 20 --graph 0 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/kt_list.json 2> $out/kt_list.err; echo "exit=$?"
i=0
shrink() {   # keep the counter rows of the loop's kernels only (the staging copies of a bench run are tens of thousands of dispatches)
  find "$1" -name "*counter_collection.csv" | while read f; do { head -1 "$f"; grep -E "rank_loop_kernel|step_kernel|event_kernel" "$f"; } > "$f.tmp"; mv "$f.tmp" "$f"; done
  find "$1" -name "*kernel_trace.csv" -delete; find "$1" -name "*agent_info.csv" -delete
}
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  echo "== pmc pass $i ($ctrs), ranks mode"
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc_ranks/p$i -- python3 bench.py --launch-shape pools --pools 1 --steps 5 --warmup 1 --other-mode 0 --host-driver 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/pmc_ranks_p$i.json 2> $out/pmc_ranks_p$i.err; echo "exit=$?"
  echo "== pmc pass $i ($ctrs), list mode"
  shrink $out/pmc_ranks/p$i
  if [ $i -le 3 ]; then
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc_list/p$i -- python3 bench.py --mode list --steps 100 --warmup 10 --profile-steps 0 --graph 0 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/pmc_list_p$i.json 2> $out/pmc_list_p$i.err; echo "exit=$?"
  shrink $out/pmc_list/p$i
  fi
done
fi
if [[ $part == *b* ]]; then
echo "== cfg3 at 1e7 photons, no profiler, then its kernel trace"
python3 bench.py --config cfg3 --steps 5 --warmup 1 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "exit=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg3 -- python3 bench.py --config cfg3 --pools 1 --steps 5 --warmup 1 --no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0 > $out/kt_cfg3.json 2> $out/kt_cfg3.err; echo "exit=$?"
echo "== cfg5 at 1e7 photons, no profiler"
python3 bench.py --config cfg5 --steps 2 --warmup 1 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "exit=$?"
echo "== cfg5 at 1e7 photons: kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg5 -- python3 bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $out/kt_cfg5.json 2> $out/kt_cfg5.err; echo "exit=$?"
echo "== kernel trace, producers and consumers around the loop (ingest, injection, output, hot table)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_ingest -- python3 tools/profile_ingest.py > $out/ingest_timings.txt 2> $out/kt_ingest.err; echo "exit=$?"
fi
# keep what travels back small: the per-dispatch traces are not needed, the stats and counter tables are
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
du -sh $out
