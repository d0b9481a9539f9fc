"""Debug helper (GPU box): tape mode engine vs oracle pass by pass on cfg1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

frame, ph, cfg = synth.config1(n_photons=1500, n0=32, n1=32)
n = len(ph["p0"])
t = np.random.default_rng(11).random(300 * (n + 4000) + 100000)
t[3::97] = 0.0
t[10:13] = 0.0
rem = 1.0 / frame["fps"]
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=1)
e.set_hydro(frame)
e.set_photons(ph)
e.set_rng_tape(t)
e.begin_frame(1, 0.5, rem)
H = O.OracleHydro(frame)
c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)
P = O.OraclePhotons(synth.photons_to_aos(ph, O.PHOTON_DTYPE))
tn, rr, sw, opos = 0.5, rem, 1, 0
for k in range(1, 120):
    st = e.run(1)
    pos, _ = e.rng_tape_position()
    rst, tn, rr, sw = O.photon_loop(c, P, H, seed=0, time_now=tn, remaining_time=rr, max_iterations=1, iteration_base=k - 1, find_switch=sw, tape=t, tape_pos=opos)
    opos = O.photon_loop.tape_pos
    out = e.get_photons()
    loc_g, loc_o = int((out["nearest_block_index"] != -1).sum()), int((P.aos["nearest_block_index"] != -1).sum())
    same_idx = np.array_equal(out["nearest_block_index"], P.aos["nearest_block_index"])
    dt = np.nanmax(np.abs(out["time_to_scatter"] - P.aos["time_to_scatter"]) / np.maximum(np.abs(P.aos["time_to_scatter"]), 1e-300))
    print("pass %3d: gpu pos %8d sc %3d rej %2d idx %5d | oracle pos %8d (this pass: sc %d rej %d) idx %5d | located %d/%d idx equal %s, max dt err %.1e"
          % (k, pos, st.frame_scatt_cnt, st.kn_rejections, st.last_scattered_index, opos, rst.frame_scatt_cnt, rst.kn_rejections, rst.last_scattered_index, loc_g, loc_o, same_idx, dt), flush=True)
    if pos != opos or not same_idx:
        break
