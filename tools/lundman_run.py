"""The reference's global validation (Doc/mcrat_doc.tex:548-566: Lundman, Pe'er & Ryde 2014 structured jet, theta_j = 0.1,
Gamma_0 = 100, L = 3e50 erg/s, p = 4) run end to end on the device: a PLUTO-like 2-D spherical frame overwritten with the
analytic jet every hydro frame (SIMULATION_TYPE = STRUCTURED_SPHERICAL_OUTFLOW), photons injected deep below the photosphere,
propagated frame by frame (phMinMax -> slab ingest -> loop) until they stream freely, then the polarisation degree against
the viewing angle.  A stress run of every stage, and a physics sanity check: Pi ~ 0 inside the core, a few per cent around
theta_v ~ theta_j, sum(U) ~ 0.      python3 tools/lundman_run.py [photons] [frames]
MODE=fast in the environment runs every frame with MCRAT_HIP_MODE_FAST (WINDOWS refreshes per frame, default 8): the same table within its
errors is the FAST mode's global check (profiles/r02_lundman_fast.txt beside profiles/r02_lundman_exact.txt)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 420
FPS, THETA_J, LUMI = 5.0, 0.1, 3e50
FAST, WINDOWS = os.environ.get("MODE", "exact") == "fast", int(os.environ.get("WINDOWS", "8"))
# the photosphere of the flow line at angle theta is r_ph = L sigma_T / (8 pi m_p c^3 eta^3), eta = Gamma_0 / sqrt(1 + (theta/theta_j)^2p):
# 1.8e11 cm on the axis, 5e12 cm at 1.3 theta_j, beyond the 2.5e13 cm domain from 1.6 theta_j on (the manual's remark, :566).  The
# slow, dense wings cost ~1e3-1e4 scatterings per photon, so the run injects within THETA_MAX and above every flow line's saturation radius
R_INJ = float(os.environ.get("R_INJ", "3e10"))
THETA_MAX = float(os.environ.get("THETA_MAX", "0.13"))
DOM = dict(r0_domain=(1e9, 2.5e13), r1_domain=(0.0, np.pi / 2), r2_domain=(0.0, 0.0))

raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e9, 0.0), (2.5e13, np.pi / 2), (2048, 512), seed=1, log_axis0=True)
jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=LUMI, theta_j=THETA_J, p=4.0)
e = engine.Engine(synth.TWO, synth.SPHERICAL, 1, virtual_rank_photons=1000)
t0 = time.perf_counter()
e.ingest(raw, dict(r_inj=R_INJ, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=FPS, **DOM), jet)
n, w = e.inject_photons(R_INJ, 1e48, N // 2, N, "b", 0.0, THETA_MAX, FPS, seed=2014)
n_lists = n // 1000
print("injected %d photons of weight %.3e at r = %.2e cm (%d lists)" % (n, w, R_INJ, n_lists), flush=True)
time_now, events, steps, quiet = 0.0, 0, 0, 0
for k in range(FRAMES):
    mm = e.ph_minmax()
    cells, ef, _ = e.ingest(raw, dict(r_inj=R_INJ, ph_inj_switch=0, min_r=mm[0], max_r=mm[1], min_theta=mm[2], max_theta=mm[3], fps=FPS, **DOM), jet)
    if FAST:
        time_now, st = e.propagate_frame_fast(time_now, (k + 1) / FPS - time_now, 5000 + k, WINDOWS)
    else:
        time_now, st = e.propagate_frame(time_now, (k + 1) / FPS - time_now, 5000 + k)
    events += st.frame_scatt_cnt
    steps += st.photon_steps
    quiet = quiet + 1 if st.frame_scatt_cnt == 0 else 0
    if k % 10 == 0 or k == FRAMES - 1:
        mx, mn, avg, ravg = e.scatt_stats()
        print("frame %3d  t = %6.1f s  <r> = %.3e cm  cells %7d (elem_factor %d)  scatterings this frame %8d  avg per photon so far %.2f  [%.1f s]"
              % (k, time_now, ravg, cells, ef, st.frame_scatt_cnt, avg, time.perf_counter() - t0), flush=True)
    if quiet >= 30:
        break
wall = time.perf_counter() - t0
out = e.get_output()
e.close()
print("mode: %s" % ("FAST, %d windows per frame" % WINDOWS if FAST else "EXACT (1000-photon lists)"))
print("total: %d scatterings, %.3e photon-steps, %d frames in %.1f s wall (ingest + loop + statistics per frame)" % (events, steps, k + 1, wall), flush=True)

# polarisation against the viewing angle (the direction the photon finally travels in)
pz = out["p3"] / out["p0"]
theta_v = np.arccos(np.clip(pz, -1, 1))
wgt = out["weight"]
edges = np.array([0.0, 0.02, 0.04, 0.06, 0.08, 0.10, 0.12, 0.14, 0.16, 0.20, 0.26])
edges = edges[edges <= THETA_MAX + 0.05]
print("theta_v / theta_j     photons      Q/I        U/I       Pi [%]   (1 sigma ~ sqrt(2/N))")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (theta_v >= lo) & (theta_v < hi)
    if m.sum() < 100:
        continue
    q = np.sum(wgt[m] * out["s1"][m]) / np.sum(wgt[m])
    u = np.sum(wgt[m] * out["s2"][m]) / np.sum(wgt[m])
    print("  %4.2f - %4.2f      %9d   %+8.4f   %+8.4f   %6.2f    (%.2f)" % (lo / THETA_J, hi / THETA_J, m.sum(), q, u, 100 * np.hypot(q, u), 100 * np.sqrt(2.0 / m.sum())))
print("mean number of scatterings per photon: %.1f; photons that never scattered: %d" % (np.average(out["num_scatt"], weights=wgt), int((out["num_scatt"] == 0).sum())))
