"""The frame queue against one launch per frame, on bench.py's headline workload (cfg2, 1e6 photons as 1025 ragged lists):
    python tools/queue_bench.py [frames] [photons] [pools-for-the-old-shape]
Prints ms per hydro frame of ALL lists for (a) restore + begin_frame + run per frame on one pool, (b) the same on three pools / streams / host
threads (round 3's headline shape), (c) mcrat_hip_pool_run_frames with restore_each_frame on one pool (one launch for all frames)."""
import os
import sys
import threading
import time

import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mcrat_amd import engine, synth  # noqa: E402

SEED = 0x4D435261


def layout(n, rank_photons=976):
    k = max(1, int(round(n / float(rank_photons))))
    ln = np.full(k, n // k, dtype=np.int64)
    ln[: n - int(ln.sum())] += 1
    d = np.random.default_rng(SEED).integers(-40, 41, k // 2)
    ln[: 2 * (k // 2): 2] += d
    ln[1: 2 * (k // 2): 2] -= d
    return k, ln, np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)


def make_pool(frame, cfg, recs, lens, offs, lo, hi, stream=None, share=None):
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream, rng_stream=0)
    if share is not None:
        e.share_hydro(share)
    else:
        e.set_hydro(frame)
    e.pool_create(hi - lo, int(lens.max()))
    for r in range(lo, hi):
        e.pool_rank(r - lo, r)
    e.pool_set_photons(list(range(hi - lo)), [recs[int(offs[r]):int(offs[r + 1])] for r in range(lo, hi)])
    e.snapshot_photons()
    return e


RESTORE = os.environ.get("RESTORE", "1") != "0"      # RESTORE=0: the photons move on from frame to frame (every path alike)


def old_frames(e, k, seed0, rem):
    ev = 0
    for f in range(k):
        if RESTORE or f == 0:
            e.restore_photons()
        e.begin_frame(seed0 + f, 0.0, rem)
        ev += e.run(0).frame_scatt_cnt
    return ev


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    pools = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    frame, ph, cfg = synth.config2(n_photons=n, seed=SEED, nzc=64, stokes=int(os.environ.get("STOKES", "0")))
    R, lens, offs = layout(n)
    recs = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
    rem = 1.0 / frame["fps"]
    out = {}

    e = make_pool(frame, cfg, recs, lens, offs, 0, R)
    old_frames(e, 2, SEED + 1000, rem)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_a = old_frames(e, K, SEED, rem)
    e.synchronize()
    out["per_frame_launch_1pool_ms"] = (time.perf_counter() - t0) / K * 1e3
    print(out, flush=True)

    def queue(k, seed0):
        seeds = np.array([[seed0 + f] * R for f in range(k)], dtype=np.uint64)
        plan, st, keep = e.frame_plan(np.ones((k, R), dtype=np.int32), seeds, np.zeros((k, R)), np.full((k, R), rem), restore_each_frame=RESTORE)
        if not RESTORE:
            e.restore_photons()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.pool_run_plan(plan, st)
        e.synchronize()
        dt = time.perf_counter() - t0
        a = np.frombuffer(st, dtype=np.dtype([(n_, "<i8") for n_ in ("it", "ps", "sc")] + [("rest", "V%d" % (C.sizeof(engine.FrameStats) - 24))]))
        return int(a["sc"].sum()), int(a["ps"].sum()), dt
    try:
        # (QUEUE_BENCH_WARM_FRAMES: the warm-up call's frames -- tools/pmc_queue.sh asks for as many as the timed call, so that the two launches of the
        # queue kernel, which have the same grid since the workgroups are persistent, are the same work and the counters' mean is one launch's)
        queue(int(os.environ.get("QUEUE_BENCH_WARM_FRAMES", "2")), SEED + 1000)
        ev_c, ps_c, dt = queue(K, SEED)
        out["queue_1pool_ms"] = dt / K * 1e3
        out["queue_frac"] = 110.0 * ps_c / dt / 8e12
        out["events_equal"] = bool(ev_a == ev_c)
        out["events_per_s_queue"] = ev_c / dt
    except engine.McratHipError as err:          # (a build without the queue, for the A/B)
        out["queue_error"] = str(err)[:60]
    e.close()

    if pools > 1:
        engines, streams = [], []
        for p in range(pools):
            lo, hi = (p * R) // pools, ((p + 1) * R) // pools
            ts = torch.cuda.Stream()
            streams.append(ts)
            engines.append(make_pool(frame, cfg, recs, lens, offs, lo, hi, stream=ts.cuda_stream, share=engines[0] if engines else None))
        gate = threading.Barrier(pools + 1)
        tot = [0] * pools

        def drive(p):
            engines[p].bind_thread()
            old_frames(engines[p], 2, SEED + 1000, rem)
            engines[p].synchronize()
            gate.wait()
            gate.wait()
            tot[p] = old_frames(engines[p], K, SEED, rem)
            engines[p].synchronize()
        th = [threading.Thread(target=drive, args=(p,)) for p in range(pools)]
        for t in th:
            t.start()
        gate.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gate.wait()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        out["per_frame_launch_%dpools_ms" % pools] = (time.perf_counter() - t0) / K * 1e3
        for x in reversed(engines):
            x.close()
    print(out)


if __name__ == "__main__":
    main()
