"""FAST mode beside the exact mode on the same photons and frame: time per frame and the distribution gates of BASELINE.md section 4
(number of scatterings, energy spectrum, Stokes Q/U).  usage: python tools/fast_vs_exact.py [n_photons] [lumi] [windows] [stokes] [cfg]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcrat_amd import engine, synth  # noqa: E402


def run(n, lumi, windows, stokes, which):
    import torch
    mk = synth.config2 if which == "cfg2" else synth.config3
    frame, ph, cfg = mk(n_photons=n, stokes=stokes, lumi=lumi)
    rem = 1.0 / frame["fps"]
    out = {}
    for mode in ("exact", "fast"):
        e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000 if mode == "exact" else 0)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.snapshot_photons()
        best = None
        for rep in range(3):
            e.restore_photons()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "exact":
                tn, st = e.propagate_frame(0.0, rem, 4242 + rep)
            else:
                tn, st = e.propagate_frame_fast(0.0, rem, 4242 + rep, windows)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[mode] = (e.get_photons(), st, best)
        e.close()
    return frame, ph, out


def gates(ph0, a, b):
    """differences between two runs in units of their Monte-Carlo error"""
    res = {}
    w = ph0["weight"]
    for name, o in (("a", a), ("b", b)):
        ns = o["num_scatt"] - ph0["num_scatt"]
        res[name] = dict(ns_mean=ns.mean(), ns_err=ns.std() / np.sqrt(len(ns)), loge=np.log(o["p0"]).mean(), loge_err=np.log(o["p0"]).std() / np.sqrt(len(ns)),
                         q=(w * o["s1"]).sum() / w.sum(), q_err=o["s1"].std() / np.sqrt(len(ns)), u=(w * o["s2"]).sum() / w.sum(), u_err=o["s2"].std() / np.sqrt(len(ns)),
                         r=np.sqrt(o["r0"] ** 2 + o["r1"] ** 2 + o["r2"] ** 2).mean())
    z = {}
    for k in ("ns_mean", "loge", "q", "u"):
        ek = {"ns_mean": "ns_err", "loge": "loge_err", "q": "q_err", "u": "u_err"}[k]
        z[k] = (res["a"][k] - res["b"][k]) / max(1e-300, np.hypot(res["a"][ek], res["b"][ek]))
    # spectrum: chi^2 per bin over 24 log-energy bins of the scattered photons
    sel_a, sel_b = a["num_scatt"] > ph0["num_scatt"], b["num_scatt"] > ph0["num_scatt"]
    lo, hi = np.log(np.concatenate([a["p0"][sel_a], b["p0"][sel_b]])).min(), np.log(np.concatenate([a["p0"][sel_a], b["p0"][sel_b]])).max()
    ha, _ = np.histogram(np.log(a["p0"][sel_a]), bins=24, range=(lo, hi))
    hb, _ = np.histogram(np.log(b["p0"][sel_b]), bins=24, range=(lo, hi))
    ok = (ha + hb) >= 20
    chi2 = (((ha - hb) ** 2) / np.maximum(1, ha + hb))[ok].sum() / max(1, ok.sum())
    return res, z, chi2


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000
    lumi = float(sys.argv[2]) if len(sys.argv) > 2 else 3e52
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    stokes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    which = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
    frame, ph, out = run(n, lumi, windows, stokes, which)
    for mode in ("exact", "fast"):
        o, st, dt = out[mode]
        print("%-5s  %.3f ms  events %d  passes %d  steps %d  relocations %d  rej %d  -> %.3g events/s" %
              (mode, dt * 1e3, st.frame_scatt_cnt, st.iterations, st.photon_steps, st.num_photons_find_new_element, st.kn_rejections, st.frame_scatt_cnt / dt))
    res, z, chi2 = gates(ph, out["exact"][0], out["fast"][0])
    print("exact", {k: float("%.6g" % v) for k, v in res["a"].items()})
    print("fast ", {k: float("%.6g" % v) for k, v in res["b"].items()})
    print("z-scores", {k: float("%.3g" % v) for k, v in z.items()}, "spectrum chi2/bin %.3g" % chi2)
