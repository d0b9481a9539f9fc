# scratch script for GPU calls during development (the round's measurements are tools/profile_round.sh)
out=gpurun_out/final; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log | cut -c1-300
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.txt
(time timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err); echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['roofline']['frac'], d['roofline']['frac_headline'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
