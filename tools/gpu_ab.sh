out=gpurun_out/r3u; rm -rf $out; mkdir -p $out
timeout -k 10 300 python bench.py --pools 1 --host-driver 0 --steps 10 --warmup 2 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/bench_p1.json 2> $out/bench_p1.err; echo "bench p1 rc=$?"
timeout -k 10 300 python bench.py --host-driver 0 --steps 20 --warmup 3 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/bench_p3.json 2> $out/bench_p3.err; echo "bench p3 rc=$?"
python - <<'PY'
import json
for f in ('bench_p1','bench_p3'):
    d=json.loads(open('gpurun_out/r3u/%s.json'%f).read().strip().splitlines()[-1])
    r=d['roofline']
    print(f, d['value'], d['ms_per_step'], 'frac', r['frac'], 'launch ms', r['avg_launch_ms'], 'headline', r.get('frac_headline'))
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $out/tests.log | cut -c1-250
