import time, sys
sys.path.insert(0, '/root/repo')
from mcrat_amd import engine, synth
frame, ph, cfg = synth.config2(n_photons=1000000)
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
for k in range(3):
    t0 = time.perf_counter(); e.set_hydro(frame); t1 = time.perf_counter()
    e.set_photons(ph); t2 = time.perf_counter()
    out = e.get_photons(); t3 = time.perf_counter()
    print("set_hydro %.3f s  set_photons %.3f s  get_photons %.3f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
for k in range(3):
    t0 = time.perf_counter(); e.set_photons_aos(aos); t1 = time.perf_counter()
    back = e.get_photons_aos(); t2 = time.perf_counter()
    print("AoS (struct photon records): set_photons %.3f s  get_photons %.3f s" % (t1 - t0, t2 - t1), flush=True)
