"""Diagnostic: where a list's workgroup spends its frame in rank_pipe_kernel (shader-clock stamps, -DMCRAT_DIAG build; run on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import build, engine, synth  # noqa: E402

engine.LIB_PATH = os.path.join(os.path.dirname(build.LIB), "libmcrat_hip_diag.so")
lib = engine.load_library()
lib.mcrat_hip_diag_rank_stamps.restype, lib.mcrat_hip_diag_rank_stamps.argtypes = C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
lib.mcrat_hip_diag_set.restype, lib.mcrat_hip_diag_set.argtypes = C.c_int, [C.c_int]
n = int(os.environ.get("N", "1000000"))
per = int(os.environ.get("PER", "976"))
lumi = float(os.environ.get("LUMI", "3e50"))
frame, ph, cfg = synth.config2(n_photons=n, lumi=lumi)
for bits, names in ((0, ["post (leftovers, min, sort)", "walk: decision", "walker waits at A", "completion + own slot", "walker's phase 1", "walker waits at B",
                         "wave 1: draws", "wave 1: phase 1"]),
                    (512, ["loads + advance", "cell record, coords, decisions", "re-location (2 slots)", "stores + shortlist", "pairs taken", "-", "-", "-"])):
    lib.mcrat_hip_diag_set(bits)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    import time
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    e.synchronize()
    t0 = time.perf_counter()
    st = e.run(0)
    wall = (time.perf_counter() - t0) * 1e3
    rows = []
    for r in range(e.num_virtual_ranks()):
        out = (C.c_longlong * 8)()
        lib.mcrat_hip_diag_rank_stamps(e.ctx, r, out)
        rows.append(list(out))
    a = np.array(rows, dtype=np.float64)
    passes = st.iterations / len(a)
    print("set %d: frame wall %.3f ms, %d lists, %.2f passes per list; ticks per list (mean) and per pass:" % (bits, wall, len(a), passes))
    for k, nm in enumerate(names):
        print("   [%d] %-32s %10.0f  %8.1f per pass" % (k, nm, a[:, k].mean(), a[:, k].mean() / passes))
    if bits == 512:
        print("   per pair: " + ", ".join("%.0f" % (a[:, k].sum() / max(a[:, 4].sum(), 1)) for k in range(4)))
    e.close()
# set C: the lists' start and end on the constant 100 MHz clock, and the shader clock against it (no other stamps: near the product build's timing)
lib.mcrat_hip_diag_set(1024)
for n_ph, per_ in ((n, per), (n, 488), (n, 1952)):
    frame, ph, cfg = synth.config2(n_photons=n_ph, lumi=lumi)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per_)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    e.synchronize()
    import time
    t0 = time.perf_counter()
    st = e.run(0)
    wall = (time.perf_counter() - t0) * 1e3
    rows = []
    for r in range(e.num_virtual_ranks()):
        out = (C.c_longlong * 8)()
        lib.mcrat_hip_diag_rank_stamps(e.ctx, r, out)
        rows.append(list(out))
    a = np.array(rows, dtype=np.float64)
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us
    dur = end - start
    print("set C (%s, %d photons per list): wall %.3f ms; %d lists, %.2f passes per list; launch span %.1f us; a list lives %.1f us on average (min %.1f, max %.1f), "
          "%.2f us per pass; sum of lives / 512 = %.1f us; shader ticks per us %.0f" % (
              "pipe" if os.environ.get("MCRAT_HIP_RANK_PIPE", "1") != "0" else "rank_loop", per_, wall, len(a), st.iterations / len(a), end.max(), dur.mean(), dur.min(), dur.max(),
              dur.sum() / st.iterations, dur.sum() / 512.0, (a[:, 2] / np.maximum(dur, 1e-9)).mean()))
    conc = [int(np.sum((start <= t) & (end > t))) for t in np.linspace(0, end.max(), 21)[:-1]]
    print("   lists alive at 20 instants across the launch:", conc)
    e.close()
