#!/bin/bash
# A/B of step_kernel's occupancy (run on the GPU box): the product library against a -DSTEP_WAVES_PER_SIMD=4 build
# (mcrat_amd/libmcrat_hip_w4.so, built beforehand), list mode, cfg2 at 1e6 and 1e7 photons.
for n in 1000000 10000000; do
  for v in "base 768" "base 1024" "w4 768" "w4 1024" "w4 2048"; do
    set -- $v
    lib=mcrat_amd/libmcrat_hip.so; [ "$1" = w4 ] && lib=mcrat_amd/libmcrat_hip_w4.so
    MCRAT_HIP_LIB=$PWD/$lib MCRAT_HIP_STEP_BLOCKS=$2 python3 bench.py --mode list --photons $n --steps 600 --warmup 50 --other-mode 0 --no-cpu-baseline --shared-clock-rounds 0 > /tmp/ab.json 2>/tmp/ab.err
    python3 -c "
import json;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print('photons $n  lib $1  blocks $2  step %.2f us  frac %.3f  event %.2f us  ms/pass %.4f' % (r['avg_launch_ms']*1e3, r['frac'], r.get('event_kernel_avg_ms',0)*1e3, d['ms_per_step']))"
  done
done
