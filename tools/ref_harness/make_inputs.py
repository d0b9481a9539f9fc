"""Writes the inputs of the reference harness (README.md): the three small parity cases of tests/golden/make_golden.py as <case>.in files.
    python tools/ref_harness/make_inputs.py OUTDIR"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mcrat_amd import synth  # noqa: E402

MAGIC = 0x4D435248
CASES = {   # name: (factory, kwargs, time_now, passes) -- the trajectories of tests/golden/make_golden.py
    "cfg1": ("config1", dict(n_photons=600, n0=16, n1=16), 0.0, 400),
    "cfg2_stokes": ("config2", dict(n_photons=600, nzc=4, stokes=1, lumi=1e54), 1.0, 400),
    "cfg3_stokes": ("config3", dict(n_photons=600, nr=128, nth=64, lumi=1e54), 2.0, 400),
}
HYDRO_COLS = ("r0", "r1", "r2", "r0_size", "r1_size", "r2_size", "r", "theta", "v0", "v1", "v2", "dens", "dens_lab", "pres", "temp", "gamma")
PHOTON_DOUBLES = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3", "num_scatt", "weight",
                  "time_to_scatter", "total_optical_depth")


def case_inputs(name):
    fac, kw, t0, passes = CASES[name]
    frame, ph, cfg = getattr(synth, fac)(**kw)
    return frame, ph, cfg, t0, passes


def write_case(name, outdir):
    frame, ph, cfg, t0, passes = case_inputs(name)
    M, N = int(frame["num_elements"]), len(ph["p0"])
    with open(os.path.join(outdir, name + ".in"), "wb") as f:
        f.write(struct.pack("<7i", MAGIC, cfg["dimensions"], cfg["geometry"], cfg["stokes"], M, N, passes))
        d2 = frame.get("r2_domain", (0.0, 0.0))
        f.write(struct.pack("<9d", frame["fps"], t0, 1.0 / frame["fps"], *frame["r0_domain"], *frame["r1_domain"], *d2))
        for k in HYDRO_COLS:
            a = np.asarray(frame[k], dtype="<f8") if k in frame else np.zeros(M, dtype="<f8")
            assert a.size == M
            f.write(a.tobytes())
        for i in range(N):
            f.write(struct.pack("<19d", *[float(ph[k][i]) for k in PHOTON_DOUBLES]))
            t = ph["type"][i]
            f.write(struct.pack("<3i", int(ph["nearest_block_index"][i]), int(ph["recalc_properties"][i]), int(t) if isinstance(t, (int, np.integer)) else ord(t)))
    print("wrote", os.path.join(outdir, name + ".in"), "(%d cells, %d photons, %d passes)" % (M, N, passes))


FUNCS_MAGIC = 0x4D435246
# the order functions.in holds them in (harness_funcs.c): name -> columns per case (0: one value per case; kn_eps has its own length)
FUNCTION_INPUTS = (("boost_beta", 3), ("boost_p", 4), ("stokes_v", 3), ("stokes_k", 3), ("stokes_kb", 3), ("stokes_in", 4), ("xy_v", 3), ("xy_ref", 3),
                   ("muller_theta", 0), ("scatter_temp", 0), ("scatter_ph_in", 4), ("scatter_stokes_in", 4), ("kns_p0", 0), ("kns_q", 0), ("kns_u", 0),
                   ("coord_xyz", 3))


def function_inputs(n=256):
    """the function-level cases G1-G7 of SURVEY.md section 8c, from a seed: what harness_funcs.c feeds MCRaT's own functions"""
    rng = np.random.default_rng(20251226)
    m_el_c = synth.M_EL * synth.C_LIGHT
    d = {"kn_eps": np.concatenate([np.logspace(-6, 3, 73), [1e-3, np.nextafter(1e-3, 0)]])}
    beta = rng.normal(size=(n, 3))
    beta *= (rng.uniform(0, 1, n) ** 0.25 * 0.99995 / np.linalg.norm(beta, axis=1))[:, None]
    beta[0] = 0.0                                                  # no boost
    beta[1] = np.array([0, 0, np.sqrt(1 - 1e-4)])                  # gamma = 100
    p = rng.normal(size=(n, 4)) * 1e-18
    p[:, 0] = np.linalg.norm(p[:, 1:], axis=1)
    d.update(boost_beta=beta, boost_p=p)
    d.update(stokes_v=rng.normal(size=(n, 3)) * 0.4, stokes_k=rng.normal(size=(n, 3)), stokes_kb=rng.normal(size=(n, 3)),
             stokes_in=np.concatenate([np.ones((n, 1)), rng.uniform(-0.5, 0.5, (n, 3))], axis=1))
    xy_v, xy_ref = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
    xy_ref[:8] = xy_v[:8] * (1 + 1e-9 * rng.normal(size=(8, 1))) + 1e-7 * rng.normal(size=(8, 3))     # near-parallel vectors
    d.update(xy_v=xy_v, xy_ref=xy_ref, muller_theta=rng.uniform(-np.pi, np.pi, n))
    temps = np.array([1e5, 5e6, 1e7, 3e7, 1e9])
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    e = 10 ** rng.uniform(-4, 0.5, n) * m_el_c
    d.update(scatter_temp=temps[np.arange(n) % 5], scatter_ph_in=np.concatenate([e[:, None], e[:, None] * dirs], axis=1),
             scatter_stokes_in=np.concatenate([np.ones((n, 1)), rng.uniform(-0.4, 0.4, (n, 2)), np.zeros((n, 1))], axis=1))
    qu = np.array([(0.0, 0.0), (1.0, 0.0), (0.3, -0.4)])
    d.update(kns_p0=np.array([1e-4, 1e-2, 1.0, 10.0])[np.arange(n) % 4] * m_el_c, kns_q=qu[(np.arange(n) // 4) % 3, 0], kns_u=qu[(np.arange(n) // 4) % 3, 1])
    xyz = rng.normal(size=(n, 3)) * 1e12
    xyz[:, 2] = np.abs(xyz[:, 2])
    d.update(coord_xyz=xyz)
    return {k: np.ascontiguousarray(v, dtype="<f8") for k, v in d.items()}


def write_functions(outdir, n=256):
    d = function_inputs(n)
    with open(os.path.join(outdir, "functions.in"), "wb") as f:
        f.write(struct.pack("<3i", FUNCS_MAGIC, d["kn_eps"].size, n))
        f.write(d["kn_eps"].tobytes())
        for k, cols in FUNCTION_INPUTS:
            assert d[k].shape == ((n, cols) if cols else (n,)), k
            f.write(d[k].tobytes())
    print("wrote", os.path.join(outdir, "functions.in"), "(%d + %d cases)" % (d["kn_eps"].size, n))


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for c in CASES:
        write_case(c, out)
    write_functions(out)
