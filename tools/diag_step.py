"""Diagnostic: time the step kernel with parts knocked out (needs the -DMCRAT_DIAG build made by this script).
Run on the GPU box:  python tools/diag_step.py
Results are wrong with any bit set; only the HIP-event timings matter."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import build, engine, synth  # noqa: E402

diag_lib = os.path.join(os.path.dirname(build.LIB), "libmcrat_hip_diag.so")
if not (os.environ.get("MCRAT_DIAG_PREBUILT") and os.path.exists(diag_lib)):     # (built here beforehand: the library travels with the snapshot)
    build.build(force=True, extra_flags=["-DMCRAT_DIAG=1"], lib=diag_lib, objdir=os.path.join(os.path.dirname(build.LIB), "_obj_diag"))
engine.LIB_PATH = diag_lib
lib = engine.load_library()
lib.mcrat_hip_diag_set.restype, lib.mcrat_hip_diag_set.argtypes = C.c_int, [C.c_int]

n = int(os.environ.get("N", "1000000"))
frame, ph, cfg = synth.config2(n_photons=n)
names = {0: "full", 64: "slow path without the cell search", 128: "slow path without boost/tau", 192: "slow path: loads + draw only",
         256: "slow path empty (queue + barrier + loop only)", 3: "no in-cell test, no slow path"}
for bits in (0, 64, 128, 192, 256, 3):
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], profile=True, iterations_per_sync=100)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, 0.2)
    lib.mcrat_hip_diag_set(0)
    w = e.run(20)
    lib.mcrat_hip_diag_set(bits)
    st = e.run(200)
    ms = (st.step_kernel_ms - w.step_kernel_ms) / max(1, st.step_kernel_launches - w.step_kernel_launches)
    ev = (st.event_kernel_ms - w.event_kernel_ms) / max(1, st.step_kernel_launches - w.step_kernel_launches)
    print("diag bits %2d %-55s step %.1f us  event %.1f us  (%d launches)" % (bits, names.get(bits, ""), ms * 1e3, ev * 1e3,
                                                                      st.step_kernel_launches - w.step_kernel_launches), flush=True)
    e.close()
lib.mcrat_hip_diag_set(0)
