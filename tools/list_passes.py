"""How predictable is a list's number of passes? (GPU box)  Two frames of the same photons with different seeds: correlation of the lists' pass counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcrat_amd import engine, synth  # noqa: E402

frame, ph, cfg = synth.config2(n_photons=1000000)
counts = []
for seed in (1, 2, 3):
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=976)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(seed, 0.0, 1.0 / frame["fps"])
    e.run(0)
    counts.append(np.array([e.rank_stats(r).iterations for r in range(e.num_virtual_ranks())], dtype=np.float64))
    e.close()
c = np.array(counts)
print("passes per list: mean %.2f, std %.2f, min %d, max %d" % (c.mean(), c[0].std(), c.min(), c.max()))
print("correlation between seeds: %.3f %.3f" % (np.corrcoef(c[0], c[1])[0, 1], np.corrcoef(c[0], c[2])[0, 1]))
m = c.mean(axis=0)
print("mean over seeds per list: std %.2f (Poisson alone would leave %.2f)" % (m.std(), np.sqrt(c.mean() / 3)))
print("first 40 lists:", c[0][:40].astype(int))
