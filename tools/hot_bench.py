"""A/B of the event walk on a hot frame (T' >= 1e7 K in the jet core: the Maxwell-Juettner rejection sampler of electron.c:202-237
accepts one attempt in 50-300): the product library (64 attempts at a time on the walk's wavefront) against a build with the one-lane
walk (-DMCRAT_NO_WAVE_WALK=1; build it beforehand with `python tools/hot_bench.py --build`, hipcc needs no GPU).
Run on the GPU box:  python tools/hot_bench.py"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import build, synth  # noqa: E402

ALT = os.path.join(os.path.dirname(build.LIB), "libmcrat_hip_nowave.so")
if "--build" in sys.argv:
    print(build.build(force=True, extra_flags=["-DMCRAT_NO_WAVE_WALK=1"], lib=ALT, objdir=os.path.join(os.path.dirname(build.LIB), "_obj_nowave")))
    sys.exit(0)
if "--child" not in sys.argv:
    for name, lib in (("one-lane walk", ALT), ("wave-wide walk", build.LIB)):
        if not os.path.exists(lib):
            print("missing", lib)
            continue
        env = dict(os.environ, MCRAT_HIP_LIB=lib)
        print("==", name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=True)
    sys.exit(0)

from mcrat_amd import engine  # noqa: E402

for label, (frame, ph, cfg) in (("cfg2 jet at r = 1e11 cm, L = 1e54 (3e7 K)", synth.config2(n_photons=1000000, lumi=1e54, r_inj=1e11, block_side=2.5e7)),
                                ("cfg2 jet at r = 1e11 cm, L = 1e53 (1.7e7 K)", synth.config2(n_photons=1000000, lumi=1e53, r_inj=1e11, block_side=2.5e7))):
    temp = frame["temp"]
    rem = 1.0 / frame["fps"]
    for per in (1000, 0):
        e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per, iterations_per_sync=256, use_graph=per == 0)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.begin_frame(1, 0.0, rem)
        e.run(1)
        e.synchronize()
        t0 = time.perf_counter()
        st = e.run(300 if per else 3000)
        e.synchronize()
        dt = time.perf_counter() - t0
        print("%-40s %-12s cells >= 1e7 K: %4.1f %%  passes %8d  events %8d  %9.3f ms  -> %.3e events/s"
              % (label, "ranks x%d" % per if per else "one list", 100.0 * np.mean(temp >= 1e7), st.iterations, st.frame_scatt_cnt, dt * 1e3, st.frame_scatt_cnt / dt),
              flush=True)
        e.close()
