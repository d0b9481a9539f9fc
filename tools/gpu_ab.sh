out=gpurun_out/r3h5; rm -rf $out; mkdir -p $out
MCRAT_H5_KEEP_FLUSH=0 timeout -k 10 500 python bench.py --steps 10 --warmup 2 --other-mode 1 --no-cpu-baseline --shared-clock-rounds 0 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h5/bench.json').read().strip().splitlines()[-1])
r=d['pcie_inclusive']['rank_pool_driver']
for k in ('checkpoints','checkpoints_and_hdf5','checkpoints_and_hdf5_files_kept_open'):
    print(k, r.get(k,{}).get('wall_ms_total'), json.dumps(r.get(k,{}).get('ms_per_frame')))
PY
