"""Debug helper (GPU box): 3-D spherical frame as virtual ranks / one list against the oracle, pass by pass."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

n = int(os.environ.get("N", "1000"))
frame, ph, cfg = synth.config_3d(synth.SPHERICAL, n_photons=n)
rem = 1.0 / frame["fps"]
c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
H = O.OracleHydro(frame)
for per in (0, 1000):
    for passes in (1, 2, 3, 5, 20):
        e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.begin_frame(7, 0.0, rem)
        st = e.run(passes)
        out = e.get_photons()
        P = O.OraclePhotons(synth.photons_to_aos(ph, O.PHOTON_DTYPE))
        rst, rtn, rrem, _ = O.photon_loop(c, P, H, seed=7, time_now=0.0, remaining_time=rem, max_iterations=passes)
        bad = {}
        for k in ("r0", "r1", "r2", "p0", "p1", "p2", "p3", "comv_p0", "s1", "s2", "time_to_scatter", "total_optical_depth"):
            a, b = out[k], P.aos[k]
            with np.errstate(all="ignore"):
                err = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
            err[np.isnan(a) & np.isnan(b)] = 0
            bad[k] = (float(np.nanmax(err)), int(np.isnan(a).sum()), int(np.isnan(b).sum()))
        print("per %d passes %d: gpu it %d sc %d rej %d reloc %d nf %d | oracle it %d sc %d rej %d | idx equal %s ns equal %s" % (
            per, passes, st.iterations, st.frame_scatt_cnt, st.kn_rejections, st.num_photons_find_new_element, getattr(st, "not_found", -1),
            rst.iterations, rst.frame_scatt_cnt, rst.kn_rejections,
            np.array_equal(out["nearest_block_index"], P.aos["nearest_block_index"]), np.array_equal(out["num_scatt"], P.aos["num_scatt"])), flush=True)
        print("      max rel err (nan gpu, nan oracle):", {k: ("%.1e" % v[0], v[1], v[2]) for k, v in bad.items()}, flush=True)
        e.close()
