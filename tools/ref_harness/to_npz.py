"""Turns the harness' <case>.out files into the fixtures tests/test_ref_vectors.py reads (README.md).
    python tools/ref_harness/to_npz.py HARNESS_OUTDIR tests/golden"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_inputs import CASES, FUNCS_MAGIC, MAGIC, PHOTON_DOUBLES  # noqa: E402


def read_out(path):
    with open(path, "rb") as f:
        magic, n, passes, scatt, relocated, last_idx = struct.unpack("<6i", f.read(24))
        assert magic == MAGIC, "not a harness output"
        time_now, remaining = struct.unpack("<2d", f.read(16))
        rec = np.dtype([("d", "<f8", 19), ("k", "<i4", 3)])
        ph = np.frombuffer(f.read(rec.itemsize * n), dtype=rec)
        (nt,) = struct.unpack("<q", f.read(8))
        tape = np.frombuffer(f.read(8 * nt), dtype="<f8").copy()
        tail = f.read(8 * 7 + 4 * 2)                       # G10: phMinMax, phScattStats, averagePhotonEnergy on the end state
    out = {k: ph["d"][:, j].copy() for j, k in enumerate(PHOTON_DOUBLES)}
    if len(tail) == 64:
        out["reductions"] = np.array(struct.unpack("<7d", tail[:56]))      # min_r, max_r, min_theta, max_theta, avg_scatt, avg_r, avg_energy
        out["scatt_max_min"] = np.array(struct.unpack("<2i", tail[56:]), dtype=np.int64)
    out["nearest_block_index"] = ph["k"][:, 0].copy()
    out["recalc_properties"] = ph["k"][:, 1].copy()
    out["type"] = ph["k"][:, 2].astype(np.uint8)
    out["tape"] = tape
    out["stats"] = np.array([passes, scatt, relocated, last_idx], dtype=np.int64)
    out["clock"] = np.array([time_now, remaining])
    return out


def read_functions(path):
    """functions_<case>.out of harness_funcs.c -> dict of arrays (its layout is in that file's header)"""
    with open(path, "rb") as f:
        magic, dims, geom, stokes, nk, n = struct.unpack("<6i", f.read(24))
        assert magic == FUNCS_MAGIC, "not a harness_funcs output"

        def dbl(*shape):
            cnt = int(np.prod(shape))
            return np.frombuffer(f.read(8 * cnt), dtype="<f8").reshape(shape).copy()

        def ints(cnt, dtype):
            return np.frombuffer(f.read(np.dtype(dtype).itemsize * cnt), dtype=dtype).copy()
        out = {"switches": np.array([dims, geom, stokes], dtype=np.int64)}
        out["kn_sigma"] = dbl(nk)
        out["boost_out_photon"], out["boost_out_electron"] = dbl(n, 4), dbl(n, 4)
        out["stokes_out"] = dbl(n, 4)
        out["xy_x"], out["xy_y"], out["xy_phi"], out["muller_out"] = dbl(n, 3), dbl(n, 3), dbl(n), dbl(n, 4)
        out["scatter_electron"], out["scatter_ph_out"], out["scatter_stokes_out"] = dbl(n, 4), dbl(n, 4), dbl(n, 4)
        out["scatter_occurred"], out["scatter_tape_at"] = ints(n, "<i4"), ints(n + 1, "<i8")
        out["kns_theta"], out["kns_phi"] = dbl(n), dbl(n)
        out["kns_ok"], out["kns_tape_at"] = ints(n, "<i4"), ints(n + 1, "<i8")
        out["coord_out"] = dbl(n, 3)
        (nt,) = struct.unpack("<q", f.read(8))
        out["tape"] = np.frombuffer(f.read(8 * nt), dtype="<f8").copy()
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    for c in CASES:
        p = os.path.join(src, "functions_%s.out" % c)
        if os.path.exists(p):
            d = read_functions(p)
            np.savez_compressed(os.path.join(dst, "ref_functions_%s.npz" % c), **d)
            print("wrote ref_functions_%s.npz: %d uniforms" % (c, d["tape"].size))
    for c in CASES:
        p = os.path.join(src, c + ".out")
        if not os.path.exists(p):
            print("no", p)
            continue
        d = read_out(p)
        np.savez_compressed(os.path.join(dst, "ref_traj_%s.npz" % c), **d)
        print("wrote ref_traj_%s.npz: %d passes, %d scatterings, %d uniforms" % (c, d["stats"][0], d["stats"][1], d["tape"].size))
