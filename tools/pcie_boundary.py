"""What the photon list costs on its way across PCIe (mcrat_hip_set_photons / mcrat_hip_get_photons, 10^6 struct photon records = 176 MB):
the caller's array pageable, as malloc hands it out, against page-locked with mcrat_hip_register_host."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n = int(os.environ.get("N", "1000000"))
frame, ph, cfg = synth.config2(n_photons=n)
aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
e.set_hydro(frame)
for mode in ("pageable", "registered"):
    buf = aos.copy()
    if mode == "registered":
        t0 = time.perf_counter()
        e.register_host(buf)
        t_reg = (time.perf_counter() - t0) * 1e3
    ts, tg = [], []
    for k in range(6):
        t0 = time.perf_counter()
        e.set_photons_aos(buf, num_null=0)
        t1 = time.perf_counter()
        e.get_photons_aos(out=buf)
        t2 = time.perf_counter()
        if k:
            ts.append(t1 - t0); tg.append(t2 - t1)
    print("%-10s set_photons %.2f ms (%.1f GB/s)  get_photons %.2f ms (%.1f GB/s)%s" % (
        mode, min(ts) * 1e3, buf.nbytes / min(ts) / 1e9, min(tg) * 1e3, buf.nbytes / min(tg) / 1e9,
        "   (hipHostRegister once: %.1f ms)" % t_reg if mode == "registered" else ""), flush=True)
    if mode == "registered":
        e.unregister_host(buf)
e.close()
