"""Diagnostic (-DMCRAT_DIAG build): shader-clock stamps inside the event walk of rank_loop_kernel, last pass of every list of a thin cfg2 frame:
sorted shortlist -> candidate loaded and advanced -> fluid frame -> thermal electron -> singleScatter -> boost back, tau, stores -> bookkeeping."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import build, engine, synth  # noqa: E402

diag_lib = os.path.join(os.path.dirname(build.LIB), "libmcrat_hip_diag.so")
if not (os.environ.get("MCRAT_DIAG_PREBUILT") and os.path.exists(diag_lib)):
    build.build(force=True, extra_flags=["-DMCRAT_DIAG=1"], lib=diag_lib, objdir=os.path.join(os.path.dirname(build.LIB), "_obj_diag"))
engine.LIB_PATH = diag_lib
lib = engine.load_library()
lib.mcrat_hip_diag_rank_stamps.restype, lib.mcrat_hip_diag_rank_stamps.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
for lumi, stokes in ((3e50, 0), (3.6e52, 0), (3.6e52, 1)):
    frame, ph, cfg = synth.config2(n_photons=1_000_000, lumi=lumi, stokes=stokes)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    st = e.run(0 if lumi < 1e51 else 60)
    rows = []
    for r in range(e.num_virtual_ranks()):
        out = (C.c_longlong * 8)()
        lib.mcrat_hip_diag_rank_stamps(e.ctx, r, out)
        rows.append(list(out))
    a = np.array(rows, dtype=np.float64)
    ok = (np.diff(a[:, 1:8], axis=1) > 0).all(axis=1)                 # lists whose last pass scattered (every stamp of one pass)
    d = np.diff(a[ok][:, 1:8], axis=1)
    med = np.median(d, axis=0)
    names = ["candidate loads + advance", "fluid frame (+ Stokes rotation)", "thermal electron", "singleScatter", "boost back + tau + stores", "bookkeeping"]
    print("lumi %.1e stokes %d: %d lists with a complete last pass; walk after the sort, median ticks (100 MHz -> x10 ns): total %.0f" % (lumi, stokes, int(ok.sum()), med.sum()))
    for n, v in zip(names, med):
        print("    %-34s %7.0f  (%4.1f %%)" % (n, v, 100 * v / med.sum()))
    e.close()
