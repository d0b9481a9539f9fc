"""FAST mode's learnt refresh cadence against the exact loop on a dense run (>= 1e7 scatterings): mean scatterings per photon with its Monte-Carlo
error, for the exact mode, FAST with the learnt cadence (fast_windows = 0) and FAST with 8 fixed windows.  The same frozen frame for FRAMES frames."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n, frames = int(os.environ.get("N", "1000000")), int(os.environ.get("FRAMES", "28"))
frame, ph, cfg = synth.config2(n_photons=n, lumi=float(os.environ.get("LUMI", "3.6e52")))
dt = 1.0 / frame["fps"]
res = {}
extra = [int(x) for x in os.environ.get("WINDOWS", "8,128").split(",") if x]
for mode, windows in [("exact", None), ("fast-auto", 0)] + [("fast-%d" % w, w) for w in extra]:
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000 if mode == "exact" else 0)
    e.set_hydro(frame)
    e.set_photons(ph)
    t, events = 0.0, 0
    t0 = time.perf_counter()
    for f in range(frames):
        if mode == "exact":
            t, st = e.propagate_frame(t, (f + 1) * dt - t, 1000 + f)
        else:
            t, st = e.propagate_frame_fast(t, (f + 1) * dt - t, 1000 + f, windows)
        events += st.frame_scatt_cnt
    wall = time.perf_counter() - t0
    ns = np.asarray(e.get_photons()["num_scatt"]) - np.asarray(ph["num_scatt"])
    e.close()
    res[mode] = (ns.mean(), ns.std() / np.sqrt(len(ns)), events, wall)
    print("%-9s %d events  <N_scatt> = %.4f +- %.4f   %.2f s" % (mode, events, ns.mean(), ns.std() / np.sqrt(len(ns)), wall), flush=True)
ex = res["exact"]
for k in [m for m in res if m != "exact"]:
    d = res[k][0] - ex[0]
    print("%-9s - exact = %+.4f (%+.3f %%, %.1f sigma)" % (k, d, 100 * d / ex[0], abs(d) / np.hypot(res[k][1], ex[1])))
