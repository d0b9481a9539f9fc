# scratch script for GPU calls during development (the round's measurements are tools/profile_round.sh)
out=gpurun_out/r3ilp; rm -rf $out; mkdir -p $out
for v in "" _ilp; do
MCRAT_HIP_LIB=mcrat_amd/libmcrat_hip$v.so timeout -k 10 300 python bench.py --pools 1 --host-driver 0 --steps 10 --warmup 2 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/b$v.json 2> $out/b$v.err; echo "rc=$?"
python - "$out/b$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'])
PY
done
