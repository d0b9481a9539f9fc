out=gpurun_out/r3m; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -25 $out/tests.log
B="--no-cpu-baseline --host-driver 0 --shared-clock-rounds 0"
timeout -k 10 300 python bench.py $B > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3m/bench.json').read().strip().splitlines()[-1])
print("%.3e ev/s"%d['value'], "ms/step %.3f"%d['ms_per_step'], {k:d['roofline'][k] for k in ('frac','frac_headline','avg_launch_ms')})
print(json.dumps(d['rank_photons_sweep'])[:1500])
print(json.dumps(d['other_mode']['roofline'])[:800])
PY
