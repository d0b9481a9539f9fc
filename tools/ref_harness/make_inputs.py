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


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for c in CASES:
        write_case(c, out)
