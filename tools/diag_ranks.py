"""Diagnostic: where a virtual rank's workgroup spends a frame (s_memtime ticks, -DMCRAT_DIAG build)."""
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
n = int(os.environ.get("N", "1000000"))
per = int(os.environ.get("PER", "1000"))
for lumi in [float(x) for x in os.environ.get("LUMI", "3e50,1e53").split(",")]:
    frame, ph, cfg = synth.config2(n_photons=n, lumi=lumi)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    import time
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    e.synchronize()
    t0 = time.perf_counter()
    st = e.run(0)
    print("frame wall time %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    rows = []
    for r in range(e.num_virtual_ranks()):
        out = (C.c_longlong * 8)()
        lib.mcrat_hip_diag_rank_stamps(e.ctx, r, out)
        rows.append(list(out))
    a = np.array(rows, dtype=np.float64) * 0.01     # us at 100 MHz
    passes = a[:, 5] * 100
    print("lumi %.0e: %d ranks x %d photons, passes per rank mean %.1f" % (lumi, len(a), per, passes.mean()), flush=True)
    print("   per workgroup [us], mean over ranks: load %.1f | forced pass step %.1f | other passes: step %.1f (%.2f per pass) | event %.1f (%.2f per pass) | store %.1f | total %.1f"
          % (a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), (a[:, 2] / np.maximum(passes - 1, 1)).mean(), a[:, 3].mean(),
             (a[:, 3] / np.maximum(passes, 1)).mean(), a[:, 4].mean(), (a[:, :5].sum(axis=1) + a[:, 6] + a[:, 7]).mean()), flush=True)
    print("   of the step time: phase 1 of thread 0's wave %.1f, barrier wait %.1f, phase 2 + minimum %.1f   (ticks x 0.01; clock ~2.1 GHz)"
          % (a[:, 6].mean(), a[:, 7].mean(), a[:, 2].mean()), flush=True)
    e.close()
