"""Diagnostic: cost of the parts of phase 1 of rank_loop_kernel, by knocking them out (-DMCRAT_DIAG build; results are wrong)."""
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
lib.mcrat_hip_diag_rank_stamps.restype, lib.mcrat_hip_diag_rank_stamps.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
lib.mcrat_hip_diag_set.restype, lib.mcrat_hip_diag_set.argtypes = C.c_int, [C.c_int]
lumi = float(os.environ.get("LUMI", "1e53"))
frame, ph, cfg = synth.config2(n_photons=1000000, lumi=lumi)
if os.environ.get("SLOW"):
    names = {0: "nothing", 64: "slow: cell search", 128: "slow: boost + tau", 256: "slow: everything"}
else:
  names = {0: "nothing", 4: "free-time sample (log)", 2: "in-cell test (cell gather)", 16: "philox", 32: "coords (sqrt)", 4 + 2 + 16 + 32: "all four"}
for bits, nme in names.items():
    lib.mcrat_hip_diag_set(bits)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    st = e.run(200)
    rows = []
    for r in range(e.num_virtual_ranks()):
        out = (C.c_longlong * 8)()
        lib.mcrat_hip_diag_rank_stamps(e.ctx, r, out)
        rows.append(list(out))
    a = np.array(rows, dtype=np.float64)
    p = np.maximum(a[:, 5] - 1, 1)
    print("knocked out %-28s forced pass %7.0f | per pass [ticks]: phase 1 %7.0f  barrier %6.0f  phase 2 + min %7.0f  event %7.0f   (passes %.0f)"
          % (nme, a[:, 1].mean(), (a[:, 6] / p).mean(), (a[:, 7] / p).mean(), (a[:, 2] / p).mean(), (a[:, 3] / (p + 1)).mean(), a[:, 5].mean()), flush=True)
    e.close()
