out=gpurun_out/r3t; rm -rf $out; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_rank_pool_host.py tests/test_gpu_output.py tests/test_gpu_cfg5_composed.py -m gpu -q -x > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $out/tests.log | cut -c1-250
[ $rc -eq 124 ] && exit 1
MCRAT_DIAG_PREBUILT=1 LUMI=3e50 PER=976 timeout -k 10 120 python tools/diag_ranks.py > $out/diag_ranks.txt 2>&1; echo "diag rc=$?"; cat $out/diag_ranks.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3t/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print(json.dumps(d.get('pcie_inclusive',{}).get('rank_pool_driver'),indent=1))
PY
