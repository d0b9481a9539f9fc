"""Diagnostic: where the event kernel's serial lane spends its time (s_memtime stamps, -DMCRAT_DIAG build)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import build, engine, synth  # noqa: E402

diag_lib = os.path.join(os.path.dirname(build.LIB), "libmcrat_hip_diag.so")
if not (os.environ.get("MCRAT_DIAG_PREBUILT") and os.path.exists(diag_lib)):     # (built here beforehand: the library travels with the snapshot)
    build.build(force=True, extra_flags=["-DMCRAT_DIAG=1"], lib=diag_lib, objdir=os.path.join(os.path.dirname(build.LIB), "_obj_diag"))
engine.LIB_PATH = diag_lib
lib = engine.load_library()
lib.mcrat_hip_diag_stamps.restype, lib.mcrat_hip_diag_stamps.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]
stokes = int(os.environ.get("STOKES", "0"))
frame, ph, cfg = synth.config2(n_photons=int(os.environ.get("N", "1000000")), stokes=stokes)
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=1)
e.set_hydro(frame)
e.set_photons(ph)
e.begin_frame(1, 0.0, 0.2)
e.run(20)
rows = []
for k in range(60):
    e.run(1)
    out = (C.c_longlong * 8)()
    lib.mcrat_hip_diag_stamps(e.ctx, out)
    rows.append(list(out))
a = np.array(rows, dtype=np.float64)
names = ["entry->walk (merge block minima, sort shortlist)", "walk start->candidate loads issued", "loads + fluid velocity + stokes pre-rotation",
         "thermal electron", "singleScatter", "boost back + stores", "bookkeeping"]
d = np.diff(a, axis=1)
ok = (d >= 0).all(axis=1) & (a[:, 1:] > 0).all(axis=1)
print("passes with a plain accept on the first candidate: %d of %d; s_memtime shader-clock ticks" % (ok.sum(), len(a)))
for j, nme in enumerate(names):
    print("  %-55s median %8.0f ticks = %6.2f us" % (nme, np.median(d[ok, j]), np.median(d[ok, j]) / 2280.0))
print("  %-55s median %8.0f ticks = %6.2f us" % ("total entry->end", np.median(a[ok, 7] - a[ok, 0]), np.median(a[ok, 7] - a[ok, 0]) / 2280.0))
