out=gpurun_out/r3l; rm -rf $out; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -6 $out/tests.log
B="--no-cpu-baseline --host-driver 0 --other-mode 0 --shared-clock-rounds 0"
timeout -k 10 120 python bench.py --pools 1 $B > $out/bench_p1.json 2> $out/bench_p1.err; echo "bench p1 rc=$?"
timeout -k 10 120 python bench.py $B > $out/bench_p3.json 2> $out/bench_p3.err; echo "bench p3 rc=$?"
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python bench.py --pools 4 $B > $out/bench_p4q8.json 2> $out/bench_p4.err; echo "bench p4 rc=$?"
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python bench.py --pools 6 $B > $out/bench_p6q8.json 2> $out/bench_p6.err; echo "bench p6 rc=$?"
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python bench.py --pools 3 $B > $out/bench_p3q8.json 2> $out/bench_p3q8.err; echo "bench p3q8 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3e ev/s"%d['value'], "ms/step %.3f"%d['ms_per_step'], "frac %.3f"%d['roofline']['frac'], "launch %.3f"%d['roofline']['avg_launch_ms'])
    except Exception as e: print(f, e)
PY
timeout -k 10 200 python tools/diag_pipe.py > $out/diag_old.txt 2>&1; echo "diag rc=$?"; grep -A1 "set C" $out/diag_old.txt
timeout -k 10 200 python tools/list_passes.py > $out/passes.txt 2>&1; echo "passes rc=$?"; cat $out/passes.txt
