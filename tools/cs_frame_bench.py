"""Times one scatter frame with CYCLOSYNCHROTRON_SWITCH on (mcrat.c:706-878) for a rank-sized list.
Run on the GPU box:  python tools/cs_frame_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n = int(os.environ.get("N", "3000"))
frame, ph, cfg = synth.config2(n_photons=n, nzc=8, lumi=3e53)
aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
nulls = np.zeros(n, dtype=engine.PHOTON_DTYPE)
nulls["type"], nulls["nearest_block_index"] = b"N", -1
both = np.concatenate([aos, nulls])
e = engine.Engine(cfg["dimensions"], cfg["geometry"], 1, cyclosynchrotron=1)
e.set_hydro(frame)
e.set_hydro_extras(np.ascontiguousarray(frame["dens"]))
for rep in range(3):
    e.set_photons_aos(both)
    t0 = time.perf_counter()
    _, st, cnt = e.scatter_frame_cyclosynch(0.0, 0.2, 31, 1e12, 1e40, 20000, 0.0, 0.05, frame["fps"], emit_pool=1, scatt_frame_number=200, inj_frame_number=200)
    dt = time.perf_counter() - t0
    out = e.get_photons_aos()
    print("%s: %d passes, %d scatterings, %d emitted, %d absorbed, %d slots, %.1f ms, %.1f us/pass, checksum %.17g"
          % ("frame %d" % rep, st.iterations, st.frame_scatt_cnt, cnt.num_cyclosynch_ph_emit, cnt.frame_abs_cnt,
             e.n, dt * 1e3, dt * 1e6 / max(st.iterations, 1), float(np.sum(out["p0"] * out["weight"]))), flush=True)
e.close()
