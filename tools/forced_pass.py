"""Duration of the forced re-location pass (step_kernel<FORCE>: every slot through the slow path) in list mode, HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n = int(os.environ.get("N", "1000000"))
for name, (frame, ph, cfg) in (("cfg2", synth.config2(n_photons=n)), ("cfg3", synth.config3(n_photons=n))):
    best = None
    for k in range(4):
        e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], profile=True, iterations_per_sync=1)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.begin_frame(1 + k, 0.0, 0.2)
        st = e.run(1)
        ms = st.step_kernel_ms
        best = ms if best is None else min(best, ms)
        e.close()
    print("%s forced pass over %d slots: %.1f us  (%.3e slots/s)" % (name, n, best * 1e3, n / (best * 1e-3)), flush=True)
