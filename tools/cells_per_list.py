"""How many photons of a list share a hydro cell (the case for an LDS tile of the fluid block): the benchmark's lists on the thin and on the
120x denser cfg2 frame after one frame of the loop."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

n, per = 1000000, 976
for lumi in (3e50, 3.6e52):
    frame, ph, cfg = synth.config2(n_photons=n, lumi=lumi)
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(1, 0.0, 1.0 / frame["fps"])
    st = e.run(0)
    idx = np.asarray(e.get_photons()["nearest_block_index"])
    e.close()
    distinct, maxshare = [], []
    for r in range(0, n // per):
        c = idx[r * per:(r + 1) * per]
        c = c[c >= 0]
        u, cnt = np.unique(c, return_counts=True)
        distinct.append(len(u) / max(1, len(c)))
        maxshare.append(cnt.max() if len(cnt) else 0)
    allu = np.unique(idx[idx >= 0])
    print("L = %.1e: %d events; cells occupied by any photon %d of %d; per list of %d: distinct cells / photons = %.3f (min %.3f), most photons in one cell %.1f (max %d)"
          % (lumi, st.frame_scatt_cnt, len(allu), frame["num_elements"], per, np.mean(distinct), np.min(distinct), np.mean(maxshare), np.max(maxshare)))
