out=gpurun_out/r3x1; rm -rf $out; mkdir -p $out
for n in 250000 1000000; do for v in "" _x1; do
MCRAT_HIP_LIB=mcrat_amd/libmcrat_hip$v.so timeout -k 10 300 python bench.py --photons $n --pools 1 --host-driver 0 --steps 10 --warmup 2 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > $out/b${v}_$n.json 2> $out/b${v}_$n.err; echo "rc=$?"
python - "$out/b${v}_$n.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], 'launch', r['avg_launch_ms'], 'frac', r['frac'])
PY
done; done
