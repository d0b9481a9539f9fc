"""Whole-frame propagation of the cfg2 workload in virtual-rank mode vs one list (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n = int(os.environ.get("N", "1000000"))
lumi = float(os.environ.get("LUMI", "3e50"))
frame, ph, cfg = synth.config2(n_photons=n, lumi=lumi)
rem = 1.0 / frame["fps"]
for per in [int(x) for x in os.environ.get("PER", "1000,2000,4000,16000").split(",")]:
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, rem)
    e.run(1)                       # forced re-location pass, timed separately
    e.synchronize()
    t0 = time.perf_counter()
    st = e.run(0)
    e.synchronize()
    dt = time.perf_counter() - t0
    print("rank size %6d  ranks %5d  passes(sum) %8d  events %7d  photon-steps %.3e  frame time %.3f ms  -> %.3e events/s  %.3e photon-steps/s"
          % (per, e.num_virtual_ranks(), st.iterations, st.frame_scatt_cnt, st.photon_steps, dt * 1e3, st.frame_scatt_cnt / dt,
             st.photon_steps / dt), flush=True)
    e.close()
if os.environ.get("LIST", "1") == "1":
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=500, use_graph=True)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, rem)
    e.run(1)
    t0 = time.perf_counter()
    st = e.run(0)
    dt = time.perf_counter() - t0
    print("single list          passes %8d  events %7d  photon-steps %.3e  frame time %.3f ms  -> %.3e events/s  %.3e photon-steps/s"
          % (st.iterations, st.frame_scatt_cnt, st.photon_steps, dt * 1e3, st.frame_scatt_cnt / dt, st.photon_steps / dt), flush=True)
