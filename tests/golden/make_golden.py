"""Regenerates the committed golden vectors:  python tests/golden/make_golden.py

Provenance: the reference ships no fixtures for this path and cannot be built in this image (GSL absent), so
these vectors come from the CPU oracle (oracle/, itself pinned by the closed-form tests of
tests/test_oracle_kat.py).  They freeze the oracle's behaviour across rounds and give the GPU tests a
reference that does not depend on rebuilding the oracle.  Inputs are regenerated from seeds by mcrat_amd.synth;
only expected outputs (and the function-level inputs) are stored.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mcrat_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
M_EL, C_LIGHT = synth.M_EL, synth.C_LIGHT

TRAJECTORIES = {
    # name: (factory name, kwargs, seed, time_now, iterations)
    "traj_cfg1": ("config1", dict(n_photons=600, n0=16, n1=16), 101, 0.0, 400),
    "traj_cfg2_stokes": ("config2", dict(n_photons=600, nzc=4, stokes=1, lumi=1e54), 102, 1.0, 400),
    "traj_cfg3_stokes": ("config3", dict(n_photons=600, nr=128, nth=64, lumi=1e54), 103, 2.0, 400),
}
PHOTON_FIELDS = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2",
                 "s0", "s1", "s2", "s3", "num_scatt", "nearest_block_index", "total_optical_depth", "time_to_scatter")


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def function_vectors():
    L = O.lib()
    rng = np.random.default_rng(2025)
    out = {}
    # G1: Klein-Nishina cross section on a log grid, both sides of the 1e-3 seam
    eps = np.concatenate([np.logspace(-6, 3, 73), [1e-3, np.nextafter(1e-3, 0)]])
    out["kn_eps"] = eps
    out["kn_sigma"] = np.array([L.orc_kleinNishinaCrossSection(float(e)) for e in eps])
    # G2: Lorentz boosts, photon ('p') and electron ('e') flavour, including beta = 0 and gamma = 100
    n = 64
    beta = rng.normal(size=(n, 3))
    beta *= (rng.uniform(0, 1, n) ** 0.25 * 0.99995 / np.linalg.norm(beta, axis=1))[:, None]
    beta[0] = 0.0
    beta[1] = np.array([0, 0, np.sqrt(1 - 1e-4)])
    p = rng.normal(size=(n, 4)) * 1e-18
    p[:, 0] = np.linalg.norm(p[:, 1:], axis=1)
    res_p, res_e = np.zeros((n, 4)), np.zeros((n, 4))
    for i in range(n):
        b, q = np.ascontiguousarray(beta[i]), np.ascontiguousarray(p[i])
        L.orc_lorentzBoost(dp(b), dp(q), dp(res_p[i]), b"p")
        L.orc_lorentzBoost(dp(b), dp(q), dp(res_e[i]), b"e")
    out.update(boost_beta=beta, boost_p=p, boost_out_photon=res_p, boost_out_electron=res_e)
    # G3: stokesRotation
    v = rng.normal(size=(n, 3)) * 0.4
    k = rng.normal(size=(n, 3))
    kb = rng.normal(size=(n, 3))
    s = np.concatenate([np.ones((n, 1)), rng.uniform(-0.5, 0.5, (n, 3))], axis=1)
    s_out = s.copy()
    for i in range(n):
        L.orc_stokesRotation(dp(np.ascontiguousarray(v[i])), dp(np.ascontiguousarray(k[i])), dp(np.ascontiguousarray(kb[i])), dp(s_out[i]))
    out.update(stokes_v=v, stokes_k=k, stokes_kb=kb, stokes_in=s, stokes_out=s_out)
    # G6: singleScatter with STOKES on, electrons from the thermal sampler, event stream (seed 7, iteration i, slot 0)
    cfg = O.make_config(O.TWO, O.CYLINDRICAL, 1)
    r = O.Rng()
    L.orc_rng_init(C.byref(r), 7, 0)
    temps = np.array([1e5, 5e6, 3e7, 1e9])
    ph_in = np.zeros((n, 4)); ph_out = np.zeros((n, 4)); el = np.zeros((n, 4)); st_out = np.zeros((n, 4)); ok = np.zeros(n, dtype=np.int32)
    for i in range(n):
        L.orc_rng_set_iteration(C.byref(r), i)
        L.orc_rng_event_begin(C.byref(r), 0)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        e = 10 ** rng.uniform(-4, 0.5) * M_EL * C_LIGHT
        ph_in[i] = [e, *(e * d)]
        ph_out[i] = ph_in[i]
        st_out[i] = [1.0, *rng.uniform(-0.4, 0.4, 2), 0.0]
        s_in = st_out[i].copy()
        L.orc_singleThermalElectron(dp(el[i]), float(temps[i % 4]), dp(ph_out[i]), C.byref(r))
        ok[i] = L.orc_singleScatter(C.byref(cfg), dp(el[i].copy()), dp(ph_out[i]), dp(st_out[i]), C.byref(r))
        out.setdefault("scatter_stokes_in", np.zeros((n, 4)))[i] = s_in
    out.update(scatter_temp=temps[np.arange(n) % 4], scatter_ph_in=ph_in, scatter_electron=el, scatter_ph_out=ph_out,
               scatter_stokes_out=st_out, scatter_occurred=ok)
    return out


def trajectory(name):
    fac, kw, seed, t0, iters = TRAJECTORIES[name]
    frame, ph, cfg = getattr(synth, fac)(**kw)
    H = O.OracleHydro(frame)
    P = O.OraclePhotons(synth.photons_to_aos(ph, O.PHOTON_DTYPE))
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    st, tn, rem, _ = O.photon_loop(c, P, H, seed=seed, time_now=t0, remaining_time=1.0 / frame["fps"], max_iterations=iters)
    out = {k: np.ascontiguousarray(P.aos[k]) for k in PHOTON_FIELDS}
    out["stats"] = np.array([st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element, st.kn_rejections,
                             st.last_scattered_index], dtype=np.int64)
    out["times"] = np.array([tn, rem, st.last_time_step])
    return out


# hydro ingest: reader buffers from mcrat_amd.synth (seeded) -> the frame getHydroData would hand to the loop
INGEST = {
    "flash": (synth.TWO, synth.CYLINDRICAL, lambda: synth.flash_raw_blocks(2e9, 8, 16, 8, 1e12 - 1.6e10, seed=21),
              dict(r_inj=1e12, ph_inj_switch=0, min_r=0.995e12, max_r=1.002e12, min_theta=0.0, max_theta=0.04, fps=5.0), (3, dict(lumi=3e50, theta_j=0.1))),
    "pluto": (synth.THREE, synth.SPHERICAL, lambda: synth.pluto_raw_grid(synth.THREE, synth.SPHERICAL, (2e11, 0.0, 0.0), (3e12, 0.5, 2 * np.pi), (40, 16, 12), seed=22, log_axis0=True),
              dict(r_inj=1e12, ph_inj_switch=0, min_r=0.97e12, max_r=1.01e12, min_theta=0.01, max_theta=0.12, fps=5.0), None),
    "chombo": (synth.TWO, synth.SPHERICAL, lambda: synth.chombo_raw(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.8), (64, 32), seed=23, logr=True),
               dict(r_inj=1e12, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=5.0), (2, {})),
}
INGEST_COLUMNS = ("r0", "r1", "r2", "r0_size", "r1_size", "v0", "v1", "dens_lab", "temp", "gamma", "r", "theta")


def ingest_vectors():
    out = {}
    for name, (dims, geom, make, slab, flow) in INGEST.items():
        cfg = O.make_config(dims, geom, 0)
        cols, ef = O.hydro_ingest(cfg, make(), slab, O.outflow(flow[0], **flow[1]) if flow else None)
        out[name + "_counts"] = np.array([cols["num_elements"], ef], dtype=np.int64)
        pick = np.unique(np.linspace(0, cols["num_elements"] - 1, 400).astype(int))      # a thinned sample keeps the fixture small
        out[name + "_pick"] = pick
        for k in INGEST_COLUMNS:
            out[name + "_" + k] = cols[k][pick]
        out[name + "_sums"] = np.array([cols[k].sum() for k in INGEST_COLUMNS])
    # createHotCrossSection: a 5 x 4 table of 4000-sample Monte-Carlo integrals on an off-reference grid
    t = np.empty((5, 4))
    O.lib().orc_createHotCrossSection(t.ctypes.data_as(C.POINTER(C.c_double)), 4, 3, *HOT_GRID, 4000, 123)
    out["hot_table"] = t
    return out


HOT_GRID = (-5.0, 1.5, -2.5, 1.0)

# cyclo-synchrotron scatter frame (mcrat.c:706-878): cfg2-like frame, 300 injected photons and 300 null slots, a pool,
# 1500 passes with replacements and a list doubling, the absorption at the end
CS_FRAME = dict(n_photons=300, nzc=8, lumi=3e53, seed=31, remaining=0.2, max_photons=2000, theta_max=0.05, iterations=1500,
                b_field_calc=1, epsilon_b=0.5, rebin_e_perc=0.1, rebin_ang=0.5, rebin_ang_phi=10.0, frames=(200, 200))
CS_FIELDS = ("p0", "p1", "p2", "p3", "comv_p0", "r0", "r1", "r2", "s0", "s1", "s2", "weight", "num_scatt", "nearest_block_index")


def cs_frame_inputs():
    k = CS_FRAME
    frame, ph, cfg = synth.config2(n_photons=k["n_photons"], nzc=k["nzc"], lumi=k["lumi"])
    aos = synth.photons_to_aos(ph, O.PHOTON_DTYPE)
    L = O.lib()
    l = O.PhotonList()
    L.orc_list_init(C.byref(l))
    both = np.concatenate([aos, aos])
    assert L.orc_list_set(C.byref(l), both.ctypes.data, len(both)) == 0
    for i in range(k["n_photons"], 2 * k["n_photons"]):
        assert L.orc_list_set_null(C.byref(l), i) == 0
    return frame, cfg, l


def cs_frame_vector():
    k = CS_FRAME
    frame, cfg, l = cs_frame_inputs()
    L = O.lib()
    H = O.OracleHydro(frame)
    c = O.make_config(cfg["dimensions"], cfg["geometry"], 1)
    dens = np.ascontiguousarray(frame["dens"])
    cs = O.CS(k["b_field_calc"], k["epsilon_b"], k["rebin_e_perc"], dp(dens), None, None, None, k["frames"][0], k["frames"][1], k["rebin_ang"],
              k["rebin_ang_phi"])
    rng = O.Rng()
    L.orc_rng_init(C.byref(rng), k["seed"], 0)
    st, cnt, t = O.Stats(), O.CSCounts(), C.c_double(0.0)
    L.orc_scatter_frame_cs(C.byref(c), C.byref(cs), C.byref(l), C.byref(H.c), C.byref(rng), C.byref(t), k["remaining"], 1e12, 1e40, k["max_photons"], 0.0,
                           k["theta_max"], 1, k["iterations"], C.byref(st), C.byref(cnt))
    buf = (C.c_char * (l.list_capacity * O.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    v = np.frombuffer(buf, dtype=O.PHOTON_DTYPE).copy()
    out = {f: np.ascontiguousarray(v[f]) for f in CS_FIELDS}
    out["type"] = np.frombuffer(v["type"].tobytes(), dtype=np.uint8).copy()
    out["stats"] = np.array([st.iterations, st.frame_scatt_cnt, st.kn_rejections, cnt.num_cyclosynch_ph_emit, cnt.scatt_cyclosynch_num_ph, cnt.frame_abs_cnt,
                             cnt.rebins, cnt.error, l.list_capacity, l.num_photons, l.num_null_photons], dtype=np.int64)
    out["times"] = np.array([t.value, cnt.n_comptonized, cnt.pool_weight])
    L.orc_list_free(C.byref(l))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "cs_frame.npz"), **cs_frame_vector())
    if "--cs-only" in sys.argv:
        sys.exit(0)
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), **ingest_vectors())
    if "--ingest-only" in sys.argv:
        sys.exit(0)
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **function_vectors())
    for name in TRAJECTORIES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **trajectory(name))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
