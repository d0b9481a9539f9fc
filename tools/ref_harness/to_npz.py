"""Turns the harness' <case>.out files into the fixtures tests/test_ref_vectors.py reads (README.md).
    python tools/ref_harness/to_npz.py HARNESS_OUTDIR tests/golden"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_inputs import CASES, MAGIC, PHOTON_DOUBLES  # noqa: E402


def read_out(path):
    with open(path, "rb") as f:
        magic, n, passes, scatt, relocated, last_idx = struct.unpack("<6i", f.read(24))
        assert magic == MAGIC, "not a harness output"
        time_now, remaining = struct.unpack("<2d", f.read(16))
        rec = np.dtype([("d", "<f8", 19), ("k", "<i4", 3)])
        ph = np.frombuffer(f.read(rec.itemsize * n), dtype=rec)
        (nt,) = struct.unpack("<q", f.read(8))
        tape = np.frombuffer(f.read(8 * nt), dtype="<f8").copy()
    out = {k: ph["d"][:, j].copy() for j, k in enumerate(PHOTON_DOUBLES)}
    out["nearest_block_index"] = ph["k"][:, 0].copy()
    out["recalc_properties"] = ph["k"][:, 1].copy()
    out["type"] = ph["k"][:, 2].astype(np.uint8)
    out["tape"] = tape
    out["stats"] = np.array([passes, scatt, relocated, last_idx], dtype=np.int64)
    out["clock"] = np.array([time_now, remaining])
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    for c in CASES:
        p = os.path.join(src, c + ".out")
        if not os.path.exists(p):
            print("no", p)
            continue
        d = read_out(p)
        np.savez_compressed(os.path.join(dst, "ref_traj_%s.npz" % c), **d)
        print("wrote ref_traj_%s.npz: %d passes, %d scatterings, %d uniforms" % (c, d["stats"][0], d["stats"][1], d["tape"].size))
