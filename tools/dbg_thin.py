import sys, numpy as np
sys.path.insert(0, '.')
from mcrat_amd import synth, engine
from oracle import oracle_py as O
frame, ph, cfg = synth.config2(n_photons=2000, nzc=8)
seed, t0, rem = 0x4D435261, 3.0, 0.2
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
e.set_hydro(frame); e.set_photons(ph); e.begin_frame(seed, t0, rem)
H = O.OracleHydro(frame); P = O.OraclePhotons(synth.photons_to_aos(ph, O.PHOTON_DTYPE))
c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
tn, r, sw, base = t0, rem, 1, 0
for it in range(12):
    st = e.run(1)
    out = e.get_photons()
    s2, tn, r, sw = O.photon_loop(c, P, H, seed=seed, time_now=tn, remaining_time=r, max_iterations=1, iteration_base=base, find_switch=sw)
    base += s2.iterations
    didx = (out["nearest_block_index"] != P.aos["nearest_block_index"]).sum()
    dr = np.abs(out["r0"]-P.aos["r0"]).max()
    print(it, "gpu reloc", st.num_photons_find_new_element, "orc reloc(this it)", s2.num_photons_find_new_element, "dt", st.last_time_step, s2.last_time_step, "idx diff", didx, "dr", dr, "rem", st.remaining_time, r)
    if r <= 0: break
