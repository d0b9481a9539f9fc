"""Times the producer / consumer kernels around the loop at BASELINE sizes (run under rocprofv3 --kernel-trace --stats for the
per-kernel table committed as profiles/r01_ingest_kernel_stats.csv):  python3 tools/profile_ingest.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

DOM = dict(r0_domain=(0.0, 5e12), r1_domain=(0.0, 2.5e13), r2_domain=(0.0, 7.0))
REPS = 5


def timed(label, fn):
    fn()
    t0 = time.perf_counter()
    for _ in range(REPS):
        out = fn()
    print("%-34s %8.3f ms   %s" % (label, (time.perf_counter() - t0) * 1e3 / REPS, out), flush=True)


def main():
    jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=3e50, theta_j=0.1)
    # cfg2: FLASH checkpoint, 16 384 leaf blocks
    side = 2.5e8
    raw = synth.flash_raw_blocks(side, 64, 128, 64, 1e12 - 64 * side, seed=1)
    e = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0)
    inj = dict(r_inj=1e12, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=5.0, **DOM)
    slab = dict(r_inj=1e12, ph_inj_switch=0, min_r=0.9985e12, max_r=1.0015e12, min_theta=0.0, max_theta=0.052, fps=5.0, **DOM)
    timed("flash ingest, injection frame", lambda: e.ingest(raw, inj, jet))
    timed("flash ingest, photons' slab", lambda: e.ingest(raw, slab, jet))
    e.ingest(raw, inj, jet)
    timed("inject 1e6 photons", lambda: e.inject_photons(1e12, 1e50, 500000, 1000000, "b", 0.0, 3.0 * np.pi / 180, 5.0, 7))
    e.begin_frame(1, 0.0, 0.2)
    e.run(50)
    timed("get_output (weight != 0 columns)", lambda: len(e.get_output()["p0"]))
    timed("get_photons_range 2^20 records", lambda: len(e.get_photons_range(0, min(e.n, 1 << 20))))
    e.close()
    # the TAU_CALCULATION == TABLE input: 221 x 81 entries x 500 000 Monte-Carlo samples
    e = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0)
    timed("hot cross-section table (reference size)", lambda: e.create_hot_cross_section().shape)
    e.close()
    # cfg3: PLUTO 2048 x 512 spherical
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e9, 0.0), (2.5e13, np.pi / 2), (2048, 512), seed=2, log_axis0=True)
    e = engine.Engine(synth.TWO, synth.SPHERICAL, 1)
    timed("pluto ingest 2048x512, inj. frame", lambda: e.ingest(raw, inj, jet))
    timed("pluto ingest 2048x512, slab", lambda: e.ingest(raw, dict(slab, max_theta=0.105), jet))
    e.close()
    # cfg5-like: PLUTO-Chombo 3-D spherical, 3 levels
    raw = synth.chombo_raw(synth.THREE, synth.SPHERICAL, (2e11, 0.0, 0.0), (3e12, 0.8, 2 * np.pi), (128, 64, 64), seed=3, logr=True)
    cells = sum(len(lv["data"]) for lv in raw["levels"]) // len(raw["var_names"])
    print("chombo: %d cells in %s boxes" % (cells, [len(lv["boxes"]) for lv in raw["levels"]]), flush=True)
    e = engine.Engine(synth.THREE, synth.SPHERICAL, 1)
    timed("chombo ingest 3-D, injection frame", lambda: e.ingest(raw, inj, None))
    timed("chombo ingest 3-D, slab", lambda: e.ingest(raw, dict(slab, min_theta=0.0, max_theta=0.2), None))
    e.close()
    # a scatter frame with CYCLOSYNCHROTRON_SWITCH on (mcrat.c:706-878): a rank-sized list (3000 injected photons + room for the pool)
    frame, ph, cfg = synth.config2(n_photons=3000, nzc=8, lumi=3e53)
    aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
    nulls = np.zeros(3000, dtype=engine.PHOTON_DTYPE)
    nulls["type"], nulls["nearest_block_index"] = b"N", -1
    both = np.concatenate([aos, nulls])
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], 1, cyclosynchrotron=1)
    e.set_hydro(frame)
    e.set_hydro_extras(np.ascontiguousarray(frame["dens"]))

    def cs_frame():
        e.set_photons_aos(both)
        t0 = time.perf_counter()
        _, st, cnt = e.scatter_frame_cyclosynch(0.0, 0.2, 31, 1e12, 1e40, 20000, 0.0, 0.05, frame["fps"], emit_pool=1, scatt_frame_number=200, inj_frame_number=200)
        dt = time.perf_counter() - t0
        return "%d passes, %d scatterings, %d emitted, %d absorbed, %d slots, %.1f us/pass" % (
            st.iterations, st.frame_scatt_cnt, cnt.num_cyclosynch_ph_emit, cnt.frame_abs_cnt, e.n, dt * 1e6 / max(st.iterations, 1))
    timed("cyclo-synchrotron frame, 3000+pool", cs_frame)
    e.close()


if __name__ == "__main__":
    main()
