"""what a drop-in caller pays around the loop, per copy-thread setting:  MCRAT_HIP_COPY_THREADS=N python3 tools/copy_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

frame, ph, cfg = synth.config2(n_photons=1_000_000)
aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
t = dict(set_hydro=0.0, set_photons=0.0, get_photons=0.0, get_output=0.0, get_soa=0.0)
reps = 5
for k in range(reps + 1):
    t0 = time.perf_counter(); e.set_hydro(frame)
    t1 = time.perf_counter(); e.set_photons_aos(aos)
    t2 = time.perf_counter(); back = e.get_photons_aos()
    t3 = time.perf_counter(); out = e.get_output()
    t4 = time.perf_counter(); soa = e.get_photons()
    t5 = time.perf_counter()
    if k:
        for name, dt in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            t[name] += dt
for k in aos.dtype.names:
    assert np.array_equal(back[k], aos[k]), k
assert np.array_equal(out["p0"], aos["p0"]) and np.array_equal(soa["r2"], aos["r2"])
print("threads", os.environ.get("MCRAT_HIP_COPY_THREADS", "default"), {k: round(v * 1e3 / reps, 2) for k, v in t.items()}, flush=True)
