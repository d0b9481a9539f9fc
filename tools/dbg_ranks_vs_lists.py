"""Debug: every virtual rank against a single-list context holding only its photons (both on the GPU)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n = int(os.environ.get("N", "6000"))
per = int(os.environ.get("PER", "1000"))
lumi = float(os.environ.get("LUMI", "1e53"))
nzc = int(os.environ.get("NZC", "64"))
frame, ph, cfg = synth.config2(n_photons=n, lumi=lumi, nzc=nzc)
rem = 1.0 / frame["fps"]
e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
e.set_hydro(frame)
e.set_photons(ph)
seed = int(os.environ.get("SEED", "7"))
split = os.environ.get("SPLIT") == "1"
e.begin_frame(seed, 0.0, rem)
if split:
    e.run(1)
e.run(0)
out = e.get_photons()
nr = e.num_virtual_ranks()
bad = 0
for r in range(nr):
    rs = e.rank_stats(r)
    lo, hi = r * per, min(n, (r + 1) * per)
    sub = {k: (v[lo:hi].copy() if hasattr(v, "__len__") and len(v) == n else v) for k, v in ph.items()}
    s = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=r)
    s.set_hydro(frame)
    s.set_photons(sub)
    s.begin_frame(seed, 0.0, rem)
    st = s.run(0)
    o = s.get_photons()
    same = all(np.array_equal(np.asarray(o[k]), np.asarray(out[k])[lo:hi], equal_nan=True) for k in ("p0", "r0", "num_scatt", "nearest_block_index", "time_to_scatter"))
    if (not same) or rs.iterations != st.iterations or r < 2: print("rank %d: iterations %d vs %d, events %d vs %d, rejections %d vs %d, photons %s"
          % (r, rs.iterations, st.iterations, rs.frame_scatt_cnt, st.frame_scatt_cnt, rs.kn_rejections, st.kn_rejections, "same" if same else "DIFFER"), flush=True)
    bad += (not same) or rs.iterations != st.iterations
    s.close()
print("mismatching ranks:", bad)
