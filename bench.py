#!/usr/bin/env python3
"""bench.py -- MCRaT's photon loop on the BASELINE.json workload (configs[1]).

    python bench.py                                   # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload: synthetic 2-D FLASH-like cylindrical GRB-jet frame (16 384 leaf blocks = 1 048 576 cells, Lundman
structured jet), 10^6 photons per GPU, Compton + Klein-Nishina, photons and frame resident in HBM before the
timed region.  Two run shapes of the same loop (Src/mcrat.c:761-851):

  --mode ranks (default)  the 10^6 photons are `--rank-photons`-sized independent photon lists ("virtual ranks"),
                          each with its own clock and RNG stream -- how MCRaT is actually run (10^3 - 5*10^3
                          photons per MPI rank, Doc/mcrat_doc.tex:165-166,222).  One workgroup runs one list's
                          whole loop in one launch.  A *step* is one hydro frame (1/fps) for all lists; every
                          step starts from the same resident photon snapshot (device-to-device restore, inside
                          the timed region) with a fresh seed.
  --mode list             ONE list of 10^6 photons with one clock: every loop pass sweeps all photons
                          (step kernel, HBM-bound) and scatters one.  A *step* is one pass.  This is the shape
                          the HBM roofline is priced on.

Prints ONE JSON line (rank 0): `value` = scatter events per second of the selected mode (whole job), plus
photon-steps/s, the roofline object of that mode's dominant kernel, a short measurement of the other mode, and
the CPU baseline (oracle/ = faithful C restatement of the reference, one host core, bounded sample, same shape).
"""

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALGORITHMIC_BYTES_PER_PHOTON_STEP = 110      # SURVEY.md section 8(d) / BASELINE.md section 2
HBM_PEAK_GBS = 8000.0                        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 0x4D435261


def committed_traffic(pattern):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<pattern>: FETCH_SIZE and WRITE_SIZE
    in separate passes, FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes).  bench.py does
    not run under the profiler, so this is the latest committed measurement of the same kernel and workload."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return d.get("traffic_bytes_per_launch"), os.path.basename(files[-1])


def sub_photons(ph, lo, hi):
    return {k: (v[lo:hi].copy() if isinstance(v, np.ndarray) else v) for k, v in ph.items()}


def cpu_baseline_list(frame, ph, cfg, n_sample, iters):
    """one list: first n_sample photons, `iters` passes timed after the forced O(N*M) re-location pass"""
    from mcrat_amd import synth
    from oracle import oracle_py as O
    H = O.OracleHydro(frame)
    P = O.OraclePhotons(synth.photons_to_aos(sub_photons(ph, 0, n_sample), O.PHOTON_DTYPE))
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    rem = 1.0 / frame["fps"]
    t0 = time.perf_counter()
    st1, tn, rem, sw = O.photon_loop(c, P, H, seed=SEED, time_now=0.0, remaining_time=rem, max_iterations=1)
    t1 = time.perf_counter()
    done, scatt, base = 0, 0, 1
    while done < iters:            # the jet is thin: re-arm the frame time so that `iters` passes can be timed
        if rem <= 0:
            rem = 1.0 / frame["fps"]
        st, tn, rem, sw = O.photon_loop(c, P, H, seed=SEED, time_now=tn, remaining_time=rem,
                                        max_iterations=iters - done, iteration_base=base, find_switch=sw)
        done += st.iterations
        base += st.iterations
        scatt += st.frame_scatt_cnt
        if st.iterations == 0:
            break
    dt = time.perf_counter() - t1
    return {"value": scatt / dt, "unit": "scatter-events/s", "cores": 1, "kind": "port",
            "photon_steps_per_s": n_sample * done / dt, "first_pass_s": t1 - t0,
            "sample": ("oracle/ (faithful C restatement; the reference needs GSL and cannot be built here), 1 thread, one list of "
                       "the first %d photons on the same %d-cell frame: %d loop passes timed after the forced O(N*M) "
                       "re-location pass (first_pass_s)" % (n_sample, frame["num_elements"], done))}


def host_cores(limit=16):
    """cores this process may use, capped at the GPU box's share per GPU"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(limit, n))


def cpu_optimised_ranks(frame, ph, cfg, per, cores, seconds=8.0):
    """the honest CPU comparison: the same algorithm and arithmetic -- bit-identical results,
    tests/test_oracle_kat.py -- with the two costs a careful CPU author would remove (exact bucket grid instead of the O(n*M)
    linear cell search; only the consumed prefix of the argsort), one rank per core on all cores, ranks handed out until
    `seconds` have passed"""
    import ctypes as C
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from mcrat_amd import synth
    from oracle import oracle_py as O
    H = O.OracleHydro(frame)
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)
    t_grid = time.perf_counter()
    O.lib().orc_grid_attach(C.byref(c), C.byref(H.c))
    t_grid = time.perf_counter() - t_grid
    n_lists = min(int(ph["p0"].size) // per, 128 * cores)      # more than the cores get through in `seconds`
    # the lists are prepared before the clock starts: only the C loop (which runs outside the GIL) is timed
    lists = [O.OraclePhotons(synth.photons_to_aos(sub_photons(ph, r * per, (r + 1) * per), O.PHOTON_DTYPE)) for r in range(n_lists)]
    nxt, lock, tot = [0], threading.Lock(), [0, 0, 0]
    t0 = time.perf_counter()

    def worker(_):
        while time.perf_counter() - t0 < seconds:
            with lock:
                r = nxt[0]
                nxt[0] += 1
            if r >= n_lists:
                return
            st, _, _, _ = O.photon_loop(c, lists[r], H, seed=SEED, time_now=0.0, remaining_time=1.0 / frame["fps"], stream=r)
            with lock:
                tot[0] += st.frame_scatt_cnt; tot[1] += st.photon_steps; tot[2] += 1
    with ThreadPoolExecutor(cores) as pool:
        list(pool.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    O.lib().orc_grid_detach()
    return {"value": tot[0] / dt, "unit": "scatter-events/s", "cores": cores, "kind": "port, optimised",
            "photon_steps_per_s": tot[1] / dt, "rank_frames": tot[2], "wall_s": dt, "grid_build_s": t_grid,
            "sample": ("oracle/ with orc_config.optimised = 1 (exact bucket grid for the cell search, prefix of the argsort; results "
                       "bit-identical to the faithful port) on %d host cores, one virtual rank of %d photons per core at a time, "
                       "%d whole rank-frames in %.1f s; the per-frame grid build (%.2f s, once per hydro frame, shared by the "
                       "ranks) is not in the rate" % (cores, per, tot[2], dt, t_grid))}


def cpu_baseline_ranks(frame, ph, cfg, per, cores):
    """virtual ranks the way the reference runs them: one rank per core, all cores at once, each rank one whole frame of its
    own `per` photons, forced re-location pass included (ranks never talk: mcrat.c has no MPI call inside the loop).  The
    oracle runs outside the GIL (ctypes), so the ranks are threads of this process; the frame is shared read-only.  The sample is
    bounded in time, not in ranks: one rank per core first, and when that took well under ten seconds (frames whose photons sit in
    low-numbered cells end the reference's linear cell search early) as many further ranks per core as fit about ten."""
    from concurrent.futures import ThreadPoolExecutor
    from mcrat_amd import synth
    from oracle import oracle_py as O
    H = O.OracleHydro(frame)
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    n_have = len(ph["weight"]) // per

    def batch(first, count):
        lists = [O.OraclePhotons(synth.photons_to_aos(sub_photons(ph, r * per, (r + 1) * per), O.PHOTON_DTYPE)) for r in range(first, first + count)]

        def one(k):
            t0 = time.perf_counter()
            st, _, _, _ = O.photon_loop(c, lists[k], H, seed=SEED, time_now=0.0, remaining_time=1.0 / frame["fps"], stream=first + k)
            return st.frame_scatt_cnt, st.photon_steps, time.perf_counter() - t0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            res = list(pool.map(one, range(count)))
        return res, time.perf_counter() - t0
    res, dt = batch(0, min(cores, n_have))
    if dt < 3.0 and n_have > cores:
        more = min(n_have - cores, cores * max(1, min(256, int(10.0 / max(dt, 1e-3)))))
        res, dt = batch(cores, more)
    scatt, steps = sum(r[0] for r in res), sum(r[1] for r in res)
    return {"value": scatt / dt, "unit": "scatter-events/s", "cores": cores, "kind": "port",
            "photon_steps_per_s": steps / dt, "seconds_per_rank_frame": float(np.mean([r[2] for r in res])), "wall_s": dt,
            "rank_frames": len(res),
            "sample": ("oracle/ (faithful C restatement; the reference needs GSL and cannot be built here) on %d host cores at "
                       "once, virtual ranks of %d photons as the reference runs its MPI ranks, one per core at a time, %d rank-frames in "
                       "all: each rank one whole frame (1/fps) on the same %d-cell frame, including its forced O(n*M) re-location "
                       "pass; value = events of all ranks / wall time" % (cores, per, len(res), frame["num_elements"]))}


def bench_cfg5(args):
    """BASELINE.json configs[4] on one GPU: a 3-D PLUTO-Chombo AMR frame in spherical coordinates (three levels, ~4e6 cells read), the
    magnetic field from the simulation, cyclo-synchrotron emission and absorption, Compton scattering with Stokes parameters; ~1e7
    photons as the lists of a rank pool (~1000 photons per adopted rank).  A step is one scatter frame of mcrat.c:706-878 for all lists --
    pool emission, the loop with the replacement of scattered pool photons, rebinning, absorption -- from a resident snapshot taken
    after the injection frame's own scatter frame."""
    import ctypes as C
    import torch
    from mcrat_amd import engine, synth
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("--config cfg5 is a one-GPU line (the reference's ranks do not communicate: N GPUs run N such pools)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the photon loop")
    steps = args.steps if args.steps > 0 else 3
    warmup = args.warmup if args.warmup >= 0 else 1
    n_target = args.photons if args.photons > 0 else 10_000_000
    scale = max(1, args.nzc // 16)                     # --nzc 64: the full mesh; smaller values shrink it for rehearsals
    fps, r_inj, th_max = 5.0, 1e12, 0.12
    dom = dict(r0_domain=(1e11, 3e12), r1_domain=(0.0, 0.5), r2_domain=(0.0, 2 * np.pi))
    raw = synth.chombo_raw(synth.THREE, synth.SPHERICAL, (1e11, 0.0, 0.0), (3e12, 0.5, 2 * np.pi), (24 * scale, 12 * scale, 12 * scale), seed=5, logr=True)
    cells_read = int(sum(len(lv["data"]) for lv in raw["levels"]) // len(raw["var_names"]))
    jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=1e52, theta_j=0.1)

    def field(cols):
        r, th = np.asarray(cols["r0"]), np.asarray(cols["r1"])
        b = 3e4 * (1e12 / r) * (1.0 + 0.3 * np.sin(3 * th))
        return [np.ascontiguousarray(0.2 * b), np.ascontiguousarray(0.1 * b), np.ascontiguousarray(b)]
    R = max(1, int(round(n_target / float(args.rank_photons))))
    max_photons = 2 * args.rank_photons
    n_pools = max(1, min(int(args.pools) if args.pools > 0 else 1, R))
    t1 = 1.0 / fps

    class Pool:
        """the adopted ranks [lo, hi) as one rank pool on its own HIP stream: frame ingested, lists injected, the injection frame's own scatter
        frame done (no pool photons yet), the next frame staged, snapshot taken"""

        def __init__(self, lo, hi):
            self.lo, self.hi = lo, hi
            self.ts = torch.cuda.Stream()
            pool = self.pool = engine.Engine(synth.THREE, synth.SPHERICAL, 1, cyclosynchrotron=1, stream=self.ts.cuda_stream, profile=True)
            self.m_inj, _, _ = pool.ingest(raw, dict(r_inj=r_inj, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=fps, **dom), jet)
            pool.pool_create(hi - lo, 4 * max_photons)
            for r in range(lo, hi):
                pool.pool_rank(r - lo, r)
            got = pool.pool_inject_photons(fps, [dict(r_inj=r_inj, ph_weight=1e50, min_photons=args.rank_photons * 3 // 4, max_photons=args.rank_photons * 3 // 2,
                                                      spect="b", theta_min=0.0, theta_max=th_max, seed=SEED + r) for r in range(lo, hi)])
            self.n = sum(g[0] for g in got)
            self.stage(0)
            pool.pool_scatter_frames_cyclosynch(self.frame_args(0, 0.0, SEED, 0), max_photons, fps, b_field_calc=2, rebin_ang_phi=45.0)
            self.m_cells = self.stage(1)
            pool.snapshot_photons()

        def stage(self, F):
            pool = self.pool
            mm = pool.ph_minmax()
            lo = min(mm[0], r_inj + synth.C_LIGHT * (F / fps - 0.5 / fps))
            hi = max(mm[1], r_inj + synth.C_LIGHT * (F / fps + 0.5 / fps))
            m, _, _ = pool.ingest(raw, dict(r_inj=r_inj, ph_inj_switch=0, min_r=lo, max_r=hi, min_theta=mm[2], max_theta=mm[3], fps=fps, **dom), jet)
            pool.set_hydro_extras(None, *field(pool.get_hydro()))
            return m

        def frame_args(self, F, t_now, seed0, emit):
            return [dict(seed=seed0 + r, time_now=t_now, remaining_time=(F + 1) / fps - t_now, r_inj=r_inj, ph_weight_suggest=1e50, theta_min=0.0, theta_max=th_max,
                         emit_pool=emit, scatt_frame_number=F, inj_frame_number=0) for r in range(self.lo, self.hi)]

        def one(self, seed0):
            pool, R_ = self.pool, self.hi - self.lo
            pool.restore_photons()
            st, cn = pool.pool_scatter_frames_cyclosynch(self.frame_args(1, t1, seed0, 1), max_photons, fps, b_field_calc=2, rebin_ang_phi=45.0)
            stride = max(1, R_ // 64)
            slots = sum(int(pool.lib.mcrat_hip_num_photon_slots(pool.pool_rank(r, self.lo + r).ctx)) for r in range(0, R_, stride)) * stride
            return (sum(x.frame_scatt_cnt for x in st), sum(x.photon_steps for x in st), sum(x.iterations for x in st), sum(x.num_cyclosynch_ph_emit for x in cn),
                    sum(x.frame_abs_cnt for x in cn), sum(x.rebins for x in cn), slots, sum(x.slot_steps for x in st))
    t0 = time.perf_counter()
    pools = [Pool((p * R) // n_pools, ((p + 1) * R) // n_pools) for p in range(n_pools)]
    setup_inject = time.perf_counter() - t0
    n, m_cells = sum(P.n for P in pools), max(P.m_cells for P in pools)
    pool = pools[0].pool
    # the pools run their frames side by side, a host thread each (the ranks are asynchronous in the reference too): the twenty-odd launches of
    # a frame each end with the lists that park last, and another pool's lists fill the device meanwhile
    import threading
    gate = threading.Barrier(n_pools + 1)
    parts = [None] * n_pools
    loop_prof = [None] * n_pools

    def drive(p):
        for k in range(warmup):
            pools[p].one(SEED + 100000 * (k + 1))
        pools[p].pool.synchronize()
        gate.wait()
        gate.wait()
        acc = [0] * 8
        ms0, l0 = pools[p].pool.profile_totals()
        for k in range(steps):
            acc = [a + b for a, b in zip(acc, pools[p].one(SEED + 7 + 1000 * k))]
        pools[p].pool.synchronize()
        ms1, l1 = pools[p].pool.profile_totals()
        parts[p] = acc
        loop_prof[p] = (ms1 - ms0, l1 - l0)
    threads = [threading.Thread(target=drive, args=(p,)) for p in range(n_pools)]
    for th in threads:
        th.start()
    gate.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gate.wait()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = [sum(x[j] for x in parts) for j in range(8)]
    # The roofline counts the slots actually taken through a pass (tot[7], mcrat_hip_frame_stats.slot_steps): a list that has doubled
    # (photons.c:112-121) is half settled null slots behind its last photon, which rank_loop_kernel's passes leave out (no cell, never a candidate,
    # time_to_scatter = 1e12/c every time) although the reference's loops walk them (mclib.c:620,684) -- iterations x list_capacity (tot[1], the
    # reference's definition of a photon-step) would credit 110 B to slots that move no byte.  Both are reported.
    achieved = ALGORITHMIC_BYTES_PER_PHOTON_STEP * tot[7] / dt / 1e9
    loop_ms, loop_launches = max(x[0] for x in loop_prof), sum(x[1] for x in loop_prof)     # (pools side by side: the longest pool's loop time)
    loop_gbs = ALGORITHMIC_BYTES_PER_PHOTON_STEP * tot[7] / (loop_ms * 1e-3) / 1e9 if loop_ms > 0 else 0.0
    loop_gbs_ref_def = ALGORITHMIC_BYTES_PER_PHOTON_STEP * tot[1] / (loop_ms * 1e-3) / 1e9 if loop_ms > 0 else 0.0
    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cfg5_cpu_baseline(pool, field, pools[0].hi - pools[0].lo, max_photons, fps, r_inj, th_max, host_cores())
        except Exception as ex:
            cpu = {"error": "%s: %s" % (type(ex).__name__, ex)}
    for P in pools:
        P.pool.close()
    out = {"metric": "photon-scatter-events/sec at 1e7 photons, 3D PLUTO-Chombo MHD jet with cyclo-synchrotron emission/absorption",
           "value": tot[0] / dt, "unit": "scatter-events/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3 / steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[4] on one GPU: 3-D PLUTO-Chombo AMR frame in spherical coordinates (%d cells read, %d in the photons' "
                                  "slab), B_FIELD_CALC == SIMULATION, cyclo-synchrotron emission and absorption, Compton+KN, Stokes on; %d injected photons "
                                  "as %d adopted ranks (lists of %d-%d photons that grow with their pool photons) in %d rank pool(s), each on its own HIP stream "
                                  "with a host thread; step = one scatter frame (mcrat.c:706-878: pool emission, loop with replacement of scattered pool "
                                  "photons, rebinning where a list is due -- %s in this run --, absorption) for all lists, from resident snapshots"
                                  % (cells_read, m_cells, n, R, args.rank_photons * 3 // 4, args.rank_photons * 3 // 2, n_pools,
                                     ("%.1f lists per frame" % (tot[5] / steps)) if tot[5] else "none"),
                      "mode": "ranks", "photons_per_gpu": n, "cells": int(m_cells), "parallelism": "independent photon shards x1"},
           "photon_steps_per_s": tot[1] / dt, "slot_steps_per_s": tot[7] / dt, "photon_steps_reference_definition": tot[1], "slot_steps_taken_through_a_pass": tot[7],
           "scatter_events": tot[0], "loop_passes": tot[2],
           "cyclosynchrotron": {"pool_photons_emitted_per_frame": tot[3] / steps, "photons_absorbed_per_frame": tot[4] / steps, "rebinnings_per_frame": tot[5] / steps,
                                "list_slots_after_a_frame": tot[6] // steps, "setup_s_ingest_and_injection": setup_inject},
           "roofline": {"kernel": "rank_loop_kernel (CSH build: the hook of mcrat.c:786-808 inside the loop)", "bound": "hbm",
                        "achieved": loop_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": loop_gbs / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_ms": loop_ms / max(1, loop_launches), "launches": int(loop_launches), "loop_ms_per_frame": loop_ms / steps,
                        "frac_whole_frame": achieved / HBM_PEAK_GBS, "achieved_whole_frame": achieved,
                        "frac_reference_photon_step_definition": loop_gbs_ref_def / HBM_PEAK_GBS,
                        "note": "110 B x the slots actually taken through a pass (slot_steps; frac_reference_photon_step_definition prices iterations x "
                                "list_capacity instead, which counts the settled null slots of doubled lists that the kernel skips: about twice the figure, and "
                                "not bytes that move).  Loop only: over the summed duration of the loop kernel's launches (HIP events on the pool's stream "
                                "around every batch of launches; the pools' launches do not overlap here when --pools is 1); frac_whole_frame divides the "
                                "same bytes by the WALL time of the timed frames -- pool emission, rebinning, absorption and the host's part included"},
           "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)


def cfg5_cpu_baseline(pool, field, R, max_photons, fps, r_inj, th_max, cores):
    """the oracle's cyclo-synchrotron scatter frame (faithful port, one list per host core, all cores at once) on the frame the device holds and on
    the device's own lists, 150 loop passes each after the pool emission"""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from mcrat_amd import synth
    from oracle import oracle_py as O
    cols = pool.get_hydro()
    Bf = field(cols)
    dens = np.ascontiguousarray(cols["dens"])
    dom = dict(r0_domain=(1e11, 3e12), r1_domain=(0.0, 0.5), r2_domain=(0.0, 2 * np.pi))
    H = O.OracleHydro(dict(cols, **dom, fps=fps))
    c = O.make_config(synth.THREE, synth.SPHERICAL, 1)
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    n_sample = min(R, 64 * cores)                    # ~10 s of CPU work on the box's cores: whole frames of 64 lists per core
    lists = [pool.pool_rank(r, r).get_photons_aos().astype(O.PHOTON_DTYPE) for r in range(n_sample)]
    L = O.lib()

    def one(r):
        cs = O.CS(2, 0.5, 0.1, ptr(dens), ptr(Bf[0]), ptr(Bf[1]), ptr(Bf[2]), 1, 0, 0.5, 45.0)
        l = O.PhotonList()
        L.orc_list_init(C.byref(l))
        a = lists[r]
        nulls = np.flatnonzero(a["type"] == b"N")
        full = a.copy()
        if len(nulls):
            full[nulls] = a[np.flatnonzero(a["type"] != b"N")[0]]
        L.orc_list_set(C.byref(l), full.ctypes.data, len(full))
        for i in nulls:
            L.orc_list_set_null(C.byref(l), int(i))
        rng = O.Rng()
        L.orc_rng_init(C.byref(rng), SEED + r, r)
        st, cnt, t = O.Stats(), O.CSCounts(), C.c_double(1.0 / fps)
        t0 = time.perf_counter()
        L.orc_scatter_frame_cs(C.byref(c), C.byref(cs), C.byref(l), C.byref(H.c), C.byref(rng), C.byref(t), 1.0 / fps, r_inj, 1e50, max_photons, 0.0, th_max, 1, 0,
                               C.byref(st), C.byref(cnt))
        dt = time.perf_counter() - t0
        L.orc_list_free(C.byref(l))
        return st.frame_scatt_cnt, st.photon_steps, dt
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, range(len(lists))))
    dt = time.perf_counter() - t0
    return {"value": sum(x[0] for x in res) / dt, "unit": "scatter-events/s", "cores": min(cores, len(lists)), "kind": "port",
            "photon_steps_per_s": sum(x[1] for x in res) / dt, "wall_s": dt, "rank_frames": len(lists),
            "sample": "oracle/ (faithful C restatement of mcrat.c:706-878 with the switch on) on %d host cores at once, one of the device's lists per core at a "
                      "time, %d lists in all, on the frame the device holds (%d cells): each list's whole scatter frame -- pool emission, the loop with the "
                      "hook, absorption" % (min(cores, len(lists)), len(lists), int(cols["num_elements"]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", choices=("ranks", "list", "shared-clock"), default="ranks")
    ap.add_argument("--steps", type=int, default=0, help="ranks: frames (default 20); list: loop passes (default 2000)")
    ap.add_argument("--warmup", type=int, default=-1, help="ranks: frames (default 2); list: passes (default 50)")
    ap.add_argument("--photons", type=int, default=0, help="photon slots per GPU (default: 1e6 for cfg2, 1e7 for cfg3)")
    ap.add_argument("--config", choices=("cfg2", "cfg3", "cfg5"), default="cfg2",
                    help="BASELINE.json configs[1] (2D FLASH-like cylindrical jet, 1e6 photons: the headline) or configs[2] (2D PLUTO-like "
                         "spherical jet, 1e7 photons, Stokes on)")
    ap.add_argument("--rank-photons", type=int, default=976, help="mean photons per adopted rank (ranks mode); the lists differ in length")
    ap.add_argument("--host-driver", type=int, default=1, help="also time the host-C rank-pool driver with its outputs (ranks mode, 1 GPU)")
    ap.add_argument("--nzc", type=int, default=64, help="mesh scale: 64 -> 1 048 576 cells")
    ap.add_argument("--stokes", type=int, default=0)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--profile-steps", type=int, default=300, help="list-mode passes bracketed by HIP events for the roofline")
    ap.add_argument("--other-mode", type=int, default=1, help="also measure the other run shape briefly")
    ap.add_argument("--pools", type=int, default=0, help="ranks mode: rank pools (HIP streams, host threads) the lists are dealt out to; 0: 3 for "
                                                        "cfg2 / cfg3 (measured on cfg2: 0.95 ms per frame with one, 0.70 with two, 0.65 with three, 1.0 with four: HIP has "
                                                        "four hardware queues), 1 for cfg5 (measured: slower with two)")
    ap.add_argument("--launch-shape", choices=("queue", "pools"), default=None,
                    help="ranks mode: 'queue' = ONE rank pool and ONE launch for all timed frames (mcrat_hip_pool_run_frames: a list that is through frame f "
                         "starts f + 1 while others are still in f; measured 0.51 ms per frame against 0.53 with three pools); 'pools' = round 3's shape, "
                         "--pools rank pools on their own streams with a host thread each, one launch per pool and frame.  Default: queue for cfg2 "
                         "(the headline), pools for cfg3 -- its 10 246 lists run as 128-thread lists, a launch form without a queue build, so the "
                         "queue's call would run the frames one launch each on one pool (7.0 ms per frame against 6.6 with three pools)")
    ap.add_argument("--share-hydro", type=int, default=1, help="the pools read one staged copy of the hydro frame (mcrat_hip_share_hydro)")
    ap.add_argument("--fast-windows", type=int, default=0, help="FAST mode beside the exact headline: refreshes per frame (0: the context learns them from the frame before, the unbiased default; < 0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shared-clock-rounds", type=int, default=300,
                    help="also time this many rounds of the one-list-over-all-GPUs mode (0: skip)")
    args = ap.parse_args()
    steps = args.steps if args.steps > 0 else (20 if args.mode == "ranks" else 2000)
    warmup = args.warmup if args.warmup >= 0 else (2 if args.mode == "ranks" else 50)
    if args.config == "cfg5":
        return bench_cfg5(args)
    if args.photons <= 0:
        args.photons = 1_000_000 if args.config == "cfg2" else 10_000_000
    if args.photons % 2:
        raise SystemExit("--photons must be even")
    if args.config == "cfg3":
        args.stokes = 1
    if args.launch_shape is None:
        args.launch_shape = "pools" if args.config == "cfg3" else "queue"

    import torch
    from mcrat_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the photon loop")
    # rehearsal on a one-GPU box: MCRAT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo for the scalar
    # exchanges (RCCL refuses two ranks on one device); the real multi-GPU run uses nccl (= RCCL), one rank per GPU
    one_device = os.environ.get("MCRAT_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cpu" if one_device else "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # every GPU owns an independent photon set on a replica of the frame (weak scaling): the reference's ranks own
    # disjoint photons and never talk during the loop -- no data-path collective
    if args.config == "cfg3":
        frame, ph, cfg = synth.config3(n_photons=args.photons, seed=SEED + rank, nr=32 * args.nzc, nth=8 * args.nzc, stokes=1)
    else:
        frame, ph, cfg = synth.config2(n_photons=args.photons, seed=SEED + rank, nzc=args.nzc, stokes=args.stokes)
    n = int(ph["p0"].size)
    # the adopted ranks' lists (mcrat_hip_pool_*): the reference's ranks hold Poisson-sized lists (mclib.c:87-136), so the lengths
    # differ -- here by up to +-40 around the mean, adding up to n exactly
    def list_layout(rank_photons):
        k = max(1, int(round(n / float(rank_photons))))
        ln = np.full(k, n // k, dtype=np.int64)
        ln[: n - int(ln.sum())] += 1
        if k > 1 and ln.min() > 80:
            d = np.random.default_rng(SEED).integers(-40, 41, k // 2)
            ln[: 2 * (k // 2) : 2] += d
            ln[1 : 2 * (k // 2) : 2] -= d
        assert int(ln.sum()) == n and ln.min() > 0
        return k, ln, np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
    n_lists, lens, offs = list_layout(args.rank_photons)
    n_lists0, lens0, offs0 = n_lists, lens, offs
    remaining = 1.0 / frame["fps"]
    stream = torch.cuda.current_stream().cuda_stream
    first_stream = rank * 100000           # RNG streams of this GPU's virtual ranks

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def make_engine(mode, profile=False, per_sync=None, photons=None, lists=None, stream_base=None, stream=stream, share_from=None, layout=None):
        if mode == "ranks":
            # a rank pool: every list its own length, stream and clock, all lists propagated by one launch (mcrat_hip_pool_*)
            src = ph if photons is None else photons
            n_lists, lens, offs = layout if layout is not None else (n_lists0, lens0, offs0)
            lo, hi = (0, n_lists) if lists is None else lists
            sb = first_stream if stream_base is None else stream_base
            e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream,
                              rng_stream=sb, profile=profile)
            if share_from is not None:
                e.share_hydro(share_from)        # the pools are in the same hydro frame: one staged copy of it for all of them
            else:
                e.set_hydro(frame)
            e.pool_create(max(1, hi - lo), int(lens.max()))
            # the lists as the reference's struct photon records, all of them in one copy and one launch (mcrat_hip_pool_set_photons; list by list
            # it was 52 small copies per list: 53 339 __amd_rocclr_copyBuffer dispatches in round 2's trace of this command)
            for r in range(lo, hi):
                e.pool_rank(r - lo, sb + r)
            recs = synth.photons_to_aos(src, engine.PHOTON_DTYPE)
            e.pool_set_photons(list(range(hi - lo)), [recs[int(offs[r]):int(offs[r + 1])] for r in range(lo, hi)])
            return e
        e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream,
                          rng_stream=first_stream, iterations_per_sync=per_sync or 500, use_graph=bool(args.graph),
                          profile=profile)
        e.set_hydro(frame)
        e.set_photons(ph)
        return e

    def run_ranks(e, k_frames, seed0):
        """k_frames frames, each from the resident snapshot with its own seed -> (events, photon_steps, passes)"""
        ev = ps = it = 0
        for k in range(k_frames):
            e.restore_photons()
            e.begin_frame(seed0 + k, 0.0, remaining)
            st = e.run(0)
            ev += st.frame_scatt_cnt
            ps += st.photon_steps
            it += st.iterations
        return ev, ps, it

    def run_pooled(l0, l1, k_frames, k_warm, photons=None, stream_base=None, layout=None, n_pools=None):
        """the lists [l0, l1) as --pools rank pools on their own HIP streams, a host thread each; k_warm untimed frames, then k_frames frames
        timed between two barriers -> (events, photon_steps, passes, seconds)"""
        import threading
        pools = max(1, min(n_pools if n_pools else (int(args.pools) if args.pools > 0 else 3), l1 - l0))
        engines, keep = [], []
        for p in range(pools):
            lo, hi = l0 + (p * (l1 - l0)) // pools, l0 + ((p + 1) * (l1 - l0)) // pools
            ts = torch.cuda.Stream()
            keep.append(ts)
            engines.append(make_engine("ranks", photons=photons, lists=(lo, hi), stream_base=stream_base, stream=ts.cuda_stream if pools > 1 else stream,
                                       share_from=engines[0] if (engines and args.share_hydro) else None, layout=layout))
            engines[-1].snapshot_photons()
        tot = [None] * pools
        gate = threading.Barrier(pools + 1)

        def drive(p):
            torch.cuda.set_device(local_rank)
            engines[p].bind_thread()          # (HIP's current device is per host thread)
            run_ranks(engines[p], k_warm, SEED + 1000)
            engines[p].synchronize()
            gate.wait()                       # warm-up done everywhere
            gate.wait()                       # the clock is running
            tot[p] = run_ranks(engines[p], k_frames, SEED)
            engines[p].synchronize()
        threads = [threading.Thread(target=drive, args=(p,)) for p in range(pools)]
        for th in threads:
            th.start()
        gate.wait()
        sync()
        t0 = time.perf_counter()
        gate.wait()
        for th in threads:
            th.join()
        sync()
        dt = time.perf_counter() - t0
        for e in reversed(engines):           # (the pools that read engines[0]'s frame go first)
            e.close()
        return sum(x[0] for x in tot), sum(x[1] for x in tot), sum(x[2] for x in tot), dt

    def run_queued(l0, l1, k_frames, k_warm, photons=None, stream_base=None, layout=None):
        """the lists [l0, l1) as ONE rank pool, all k_frames frames in ONE launch (the frame queue): every frame of a list starts from the resident
        snapshot with the frame's seed, exactly the frames run_ranks gives it one launch at a time -> (events, photon_steps, passes, seconds, launch_ms)"""
        e = make_engine("ranks", profile=True, photons=photons, lists=(l0, l1), stream_base=stream_base, layout=layout)
        e.snapshot_photons()
        R = l1 - l0

        def plan(k, seed0):
            seeds = np.repeat(np.arange(seed0, seed0 + k, dtype=np.uint64)[:, None], R, axis=1)
            return e.frame_plan(np.ones((k, R), dtype=np.int32), seeds, np.zeros((k, R)), np.full((k, R), remaining), restore_each_frame=True)
        # (a plan of the timed call's size in which no list opens a frame: the queue's device and pinned buffers get their size outside the timed region)
        pr, sr, keep_r = e.frame_plan(np.zeros((k_frames, R), dtype=np.int32), np.zeros((k_frames, R), dtype=np.uint64), np.zeros((k_frames, R)),
                                      np.full((k_frames, R), remaining), restore_each_frame=True)
        e.pool_run_plan(pr, sr)
        if k_warm > 0:
            pw, sw, keep_w = plan(k_warm, SEED + 1000)
            e.pool_run_plan(pw, sw)
        pt, st, keep_t = plan(k_frames, SEED)
        e.synchronize()
        sync()
        t0 = time.perf_counter()
        e.pool_run_plan(pt, st)
        e.synchronize()
        sync()
        dt = time.perf_counter() - t0
        a = np.frombuffer(st, dtype=np.dtype([("it", "<i8"), ("ps", "<i8"), ("sc", "<i8"), ("rest", "V%d" % (C.sizeof(engine.FrameStats) - 24))]))
        launch_ms = (st[0].step_kernel_ms, int(st[0].step_kernel_launches))      # summed duration and number of the loop kernel's launches in the timed call
        e.close()
        return int(a["sc"].sum()), int(a["ps"].sum()), int(a["it"].sum()), dt, launch_ms

    def measure_ranks(k_frames, k_warm, with_roofline):
        # The adopted ranks never wait for each other (the reference's MPI ranks are asynchronous processes), so they need not share one
        # launch either: --pools P deals the lists out to P rank pools, each on its own HIP stream and driven by its own host thread
        # (as P processes sharing the GPU would be).  A step is still one hydro frame for ALL lists; what changes is that a pool whose last
        # lists are finishing no longer leaves the rest of the device idle -- the other pools' next frames fill it.
        # --launch-shape queue (round 4): the same independence taken further -- ONE pool, ONE launch for all k_frames frames, a workgroup per
        # (frame, list) item: a list that is through frame f starts f + 1 as soon as a workgroup slot is free (mcrat_hip_pool_run_frames).
        queue_launch_ms = None
        if args.launch_shape == "queue":
            ev, ps, it, dt, queue_launch_ms = run_queued(0, n_lists, k_frames, k_warm)
        else:
            ev, ps, it, dt = run_pooled(0, n_lists, k_frames, k_warm)
        nr = n_lists
        roof = None
        if with_roofline:
            # rank_loop_kernel between HIP events (one launch per frame here); algorithmic bytes = 110 B x the
            # photon-steps that launch performed
            p = make_engine("ranks", profile=True)
            p.snapshot_photons()
            ms = psteps = launches = 0
            for k in range(5):
                p.restore_photons()
                p.begin_frame(SEED + 5000 + k, 0.0, remaining)
                st = p.run(0)
                ms += st.step_kernel_ms
                launches += st.step_kernel_launches
                psteps += st.photon_steps
            p.close()
            launch_ms = ms / max(1, launches)
            bytes_per_launch = ALGORITHMIC_BYTES_PER_PHOTON_STEP * psteps / max(1, launches)
            achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
            full = (n == 1_000_000 and args.nzc == 64 and args.config == "cfg2")
            traffic, src = committed_traffic("*_rank_loop_kernel_pmc.json") if full else (None, None)
            # ... and the same bytes over the WALL time of the timed region (the headline's launch shape: the pools' launches overlap)
            headline_gbs = ALGORITHMIC_BYTES_PER_PHOTON_STEP * ps / dt / 1e9
            if queue_launch_ms:
                # the headline IS one launch of rank_loop_kernel (all timed frames): its duration between HIP events on the pool's stream.
                # traffic: the committed PMC passes over such a launch (tools/pmc_queue.sh), per frame x the frames of this one
                # (a launch form without a queue build -- 128- or 512-thread lists, columns in HBM/L2: cfg3's 10 246 lists -- runs the plan one launch
                # per frame inside the same call: q_launches > 1, and the figures are per launch)
                q_total_ms, q_launches = queue_launch_ms
                q_launches = max(1, q_launches)
                q_traffic, q_src = None, None
                qf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rank_loop_queue_pmc.json")))
                if qf and full:
                    with open(qf[-1]) as f:
                        q_traffic, q_src = json.load(f).get("traffic_bytes_per_frame", 0) * (k_frames // q_launches), os.path.basename(qf[-1])
                elif traffic:
                    q_traffic, q_src = traffic * (k_frames // q_launches), src
                q_bytes = ALGORITHMIC_BYTES_PER_PHOTON_STEP * ps / q_launches
                queue_launch_ms = q_total_ms / q_launches
                q_gbs = q_bytes / (queue_launch_ms * 1e-3) / 1e9
                roof = {"kernel": "rank_loop_kernel", "bound": "hbm", "achieved": q_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": q_gbs / HBM_PEAK_GBS, "frac_headline": headline_gbs / HBM_PEAK_GBS, "achieved_headline": headline_gbs,
                        "frac_one_frame_per_launch": achieved / HBM_PEAK_GBS,
                        "traffic": q_traffic, "traffic_source": q_src,
                        "bytes_per_launch": q_bytes, "avg_launch_ms": queue_launch_ms, "launches": q_launches, "frames_per_launch": k_frames // q_launches,
                        "one_frame_launch": {"bytes_per_launch": bytes_per_launch, "avg_launch_ms": launch_ms, "launches": int(launches)},
                        "note": "the timed region is ONE call for all %d frames: one launch of rank_loop_kernel (frame queue: a workgroup per (frame, list) item) "
                                "where the launch form has a queue build, else one launch per frame (`launches`); "
                                "frac = 110 B x the photon-steps of a launch / its duration between HIP events on the pool's stream / 8 TB/s; frac_headline "
                                "= the same bytes over the wall time of the timed region; frac_one_frame_per_launch = round 3's figure, one launch per frame "
                                "alone on the device (its last lists' tail inside); traffic = the committed per-frame PMC figure x frames.  A latency-bound "
                                "kernel (one workgroup walks one list's whole frame); the HBM-bound kernel of this path is step_kernel, see other_mode.roofline"
                                % k_frames}
            else:
                roof = {"kernel": "rank_loop_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "frac_kernel_alone": achieved / HBM_PEAK_GBS,
                        "frac_headline": headline_gbs / HBM_PEAK_GBS, "achieved_headline": headline_gbs,
                        "traffic": traffic, "traffic_source": src,
                        "bytes_per_launch": bytes_per_launch, "avg_launch_ms": launch_ms, "launches": int(launches),
                        "note": "latency-bound persistent kernel (one workgroup walks one list's whole frame; the forced "
                                "re-location pass of the new frame is inside the launch), measured on ONE pool holding all lists -- one "
                                "launch per frame, alone on the device (with --pools > 1 the headline's launches overlap each other, "
                                "which is the point, and have no duration of their own); the HBM-bound kernel of this path is "
                                "step_kernel, see other_mode.roofline"}
        return dict(events=ev, photon_steps=ps, passes=it, seconds=dt, ranks=nr, roofline=roof)

    def measure_strong(k_frames, k_warm):
        """strong scaling beside the weak line: ONE set of n photons (the set rank 0 holds in the weak run), its lists dealt out to the
        GPUs in contiguous blocks (sharding.shard_bounds), no data-path collective; at N = 1 this is the weak line's own workload"""
        from mcrat_amd import sharding
        if world == 1:
            common = ph
        elif args.config == "cfg3":
            _, common, _ = synth.config3(n_photons=args.photons, seed=SEED, nr=32 * args.nzc, nth=8 * args.nzc, stokes=1)
        else:
            _, common, _ = synth.config2(n_photons=args.photons, seed=SEED, nzc=args.nzc, stokes=args.stokes)
        lo, hi = sharding.shard_bounds(n_lists, world, rank)
        if args.launch_shape == "queue":
            ev, ps, it, dt, _ = run_queued(lo, hi, k_frames, k_warm, photons=common, stream_base=0)
        else:
            ev, ps, it, dt = run_pooled(lo, hi, k_frames, k_warm, photons=common, stream_base=0)
        return ev, ps, dt, hi - lo

    def measure_list(k_steps, k_warm, prof_steps):
        e = make_engine("list", per_sync=max(50, min(500, k_steps)))
        e.begin_frame(SEED, 0.0, remaining)
        w = e.run(k_warm) if k_warm > 0 else None          # includes the forced re-location pass
        it0, sc0 = (w.iterations, w.frame_scatt_cnt) if w else (0, 0)
        rel0 = w.num_photons_find_new_element if w else 0
        sync()
        t0 = time.perf_counter()
        st = e.run(k_steps)
        sync()
        dt = time.perf_counter() - t0
        if st.iterations - it0 != k_steps:
            raise SystemExit("the frame ended after %d of %d timed passes; lower --steps" % (st.iterations - it0, k_steps))
        roof = None
        if prof_steps > 0:
            # the step kernel's launches bracketed by HIP events, in a separate pass that continues the same frame
            p = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream,
                              rng_stream=first_stream, iterations_per_sync=min(500, prof_steps), profile=True)
            p.set_hydro(frame)
            p.set_photons(e.get_photons())
            p.begin_frame(SEED + 1, 0.0, remaining)
            p.step_locate_sample(0)      # the photons are already located: skip the forced pass
            p0 = p.run(20)
            ps = p.run(prof_steps)
            launches = ps.step_kernel_launches - p0.step_kernel_launches
            avg_ms = (ps.step_kernel_ms - p0.step_kernel_ms) / max(1, launches)
            ev_ms = (ps.event_kernel_ms - p0.event_kernel_ms) / max(1, launches)
            achieved = ALGORITHMIC_BYTES_PER_PHOTON_STEP * n / (avg_ms * 1e-3) / 1e9
            traffic, src = committed_traffic("*_step_kernel_pmc.json") if (n == 1_000_000 and args.nzc == 64) else (None, None)
            pass_gbs = ALGORITHMIC_BYTES_PER_PHOTON_STEP * n / (dt / k_steps) / 1e9      # the whole pass (step + event + launch gaps), wall time of the timed passes
            roof = {"kernel": "step_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                    "bytes_per_launch": ALGORITHMIC_BYTES_PER_PHOTON_STEP * n, "avg_launch_ms": avg_ms,
                    "launches": int(launches), "event_kernel_avg_ms": ev_ms,
                    "whole_pass": {"note": "the same 110 B x N over a whole loop pass -- step_kernel, then event_kernel on one workgroup, which the next "
                                           "step waits for: what a single list of N photons (and a shared-clock shard) actually gets",
                                   "ms_per_pass": dt * 1e3 / k_steps, "kernels_ms_per_pass": avg_ms + ev_ms,
                                   "achieved": pass_gbs, "frac": pass_gbs / HBM_PEAK_GBS}}
            p.close()
        e.close()
        return dict(events=st.frame_scatt_cnt - sc0, photon_steps=n * k_steps, passes=k_steps, seconds=dt, roofline=roof,
                    relocations_per_pass=(st.num_photons_find_new_element - rel0) / k_steps,
                    kn_rejections=st.kn_rejections, rescans=st.rescans)

    def measure_shared(k_rounds, k_warm):
        """ONE list of world x n photons with one clock, its slots spread over the GPUs (mcrat_amd/shared_clock.py):
        per round every GPU steps its own slots, the 736-B proposals are all-gathered (RCCL), and every GPU walks the
        merged candidates.  A round decides one loop pass unless it ends undecided (midpass, Klein-Nishina chains)."""
        from mcrat_amd import shared_clock
        e = shared_clock.make_engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, rng_stream=0)
        e.set_hydro(frame)
        e.set_photons(ph)
        sc = shared_clock.SharedClock(e, world, rank, rank * n, host_staged=(one_device and world > 1))
        e.begin_frame(SEED, 0.0, remaining)
        with torch.cuda.stream(sc.stream):
            for _ in range(k_warm):
                sc.round()                                   # includes the forced re-location pass
            _, w = e.shared_clock_poll()
            sync()
            t0 = time.perf_counter()
            for _ in range(k_rounds):
                sc.round()
            done, st = e.shared_clock_poll()
            sync()
            dt = time.perf_counter() - t0
        if done:
            raise SystemExit("the frame ended inside the timed rounds; lower the round count")
        passes = st.iterations - w.iterations
        e.close()
        return dict(events=st.frame_scatt_cnt - w.frame_scatt_cnt, photon_steps=n * passes, passes=passes, seconds=dt,
                    roofline=None, midpass_rounds=st.rescans - w.rescans, rounds=k_rounds)

    if args.mode == "ranks":
        main_res = measure_ranks(steps, warmup, rank == 0)
    elif args.mode == "list":
        main_res = measure_list(steps, warmup, args.profile_steps if rank == 0 else 0)
    else:
        main_res = measure_shared(steps, warmup)

    t_max, events_all, steps_all = main_res["seconds"], float(main_res["events"]), float(main_res["photon_steps"])
    if dist is not None:
        t = torch.tensor([t_max], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([events_all, steps_all], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t_max, events_all, steps_all = float(t.item()), float(c[0].item()), float(c[1].item())
        if args.mode == "shared-clock":
            events_all = float(main_res["events"])       # one list: every GPU counts the same events

    strong = None
    if args.mode == "ranks" and args.other_mode:
        try:
            ev_s, ps_s, dt_s, my_lists = measure_strong(steps, warmup)
            if dist is not None:
                tt = torch.tensor([dt_s], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                cc = torch.tensor([float(ev_s), float(ps_s)], dtype=torch.float64, device=red_dev)
                dist.all_reduce(cc, op=dist.ReduceOp.SUM)
                dt_s, ev_s, ps_s = float(tt.item()), float(cc[0].item()), float(cc[1].item())
            strong = {"scaling": "strong", "note": "one set of %d photons (%d lists) dealt out to the %d GPU(s) in contiguous blocks of lists, frame "
                                                   "replicated, no data-path collective; at n_gpus = 1 this is the weak line's own workload" % (n, n_lists, world),
                      "value": ev_s / dt_s, "unit": "scatter-events/s", "n_gpus": world, "photons_total": n, "lists_per_gpu": int(my_lists),
                      "ms_per_step": dt_s * 1e3 / steps, "photon_steps_per_s": ps_s / dt_s}
        except Exception as ex:
            strong = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # FAST mode beside the exact headline, never instead of it (mcrat_hip_propagate_frame_mode): the same photons and
    # frame, every photon on its own clock.  Statistically equivalent to the exact loop (tests/test_gpu_fast_mode.py), not sequence-equivalent.
    fast = None
    if args.mode == "ranks" and args.other_mode and args.fast_windows >= 0:
        try:
            def run_fast(e, k, seed0):
                ev = ps = 0
                for j in range(k):
                    e.restore_photons()
                    _, st = e.propagate_frame_fast(0.0, remaining, seed0 + j, args.fast_windows)
                    ev += st.frame_scatt_cnt
                    ps += st.photon_steps
                return ev, ps
            e = make_engine("list")
            e.snapshot_photons()
            run_fast(e, warmup, SEED + 7000)
            sync()
            t0 = time.perf_counter()
            ev_f, ps_f = run_fast(e, steps, SEED + 8000)
            sync()
            dt_f = time.perf_counter() - t0
            e.close()
            if dist is not None:
                tt = torch.tensor([dt_f], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                cc = torch.tensor([float(ev_f), float(ps_f)], dtype=torch.float64, device=red_dev)
                dist.all_reduce(cc, op=dist.ReduceOp.SUM)
                dt_f, ev_f, ps_f = float(tt.item()), float(cc[0].item()), float(cc[1].item())
            fast = {"mode": "FAST", "note": "the headline's photons and frame with MCRAT_HIP_MODE_FAST: one lane per photon through the whole frame on its "
                                            "own clock, per-photon keyed random numbers, cell and optical depth refreshed %s per frame and after each own "
                                            "scattering; statistically (not sequence-) equivalent to the exact loop: scatterings per photon, spectrum, Q/U agree "
                                            "within Monte-Carlo error (tests/test_gpu_fast_mode.py).  Reported beside the exact headline, not instead of it"
                                            % (("%d times" % args.fast_windows) if args.fast_windows > 0 else
                                               "as often as the frame before had scatterings per 1000 photons (8 ... 2048)"),
                    "windows": args.fast_windows, "value": ev_f / dt_f, "unit": "scatter-events/s", "n_gpus": world, "steps": steps,
                    "ms_per_step": dt_f * 1e3 / steps, "photon_steps_per_s": ps_f / dt_f, "scatter_events": ev_f}
            if rank == 0 and world == 1 and args.config == "cfg2":
                dframe, dph, dcfg = synth.config2(n_photons=n, seed=SEED, nzc=args.nzc, stokes=args.stokes, lumi=3.6e52)
                res = {}
                for which in ("exact", "fast"):
                    d = engine.Engine(dcfg["dimensions"], dcfg["geometry"], dcfg["stokes"], device=local_rank, stream=stream, rng_stream=first_stream,
                                      virtual_rank_photons=1000 if which == "exact" else 0)
                    d.set_hydro(dframe)
                    d.set_photons(dph)
                    d.snapshot_photons()
                    best = None
                    for j in range(3):
                        d.restore_photons()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if which == "exact":
                            _, st = d.propagate_frame(0.0, remaining, SEED + j)
                        else:
                            _, st = d.propagate_frame_fast(0.0, remaining, SEED + j, args.fast_windows)
                        torch.cuda.synchronize()
                        dtd = time.perf_counter() - t0
                        if best is None or dtd < best[0]:
                            best = (dtd, int(st.frame_scatt_cnt))
                    d.close()
                    res[which] = {"ms_per_frame": best[0] * 1e3, "scatter_events": best[1], "scatter_events_per_s": best[1] / best[0]}
                fast["dense_frame"] = {"workload": "the same mesh and photons with L = 3.6e52 erg/s (120 x denser): ~0.4 scatterings per photon and frame",
                                       "exact": res["exact"], "fast": res["fast"]}
        except Exception as ex:
            fast = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # lists of the reference's whole range of photons per rank (sample_mc.par:21-22: 1000 - 5000): the same n photons cut into lists of
    # about 1000, 2000 and 5000 -- the loop's work per event grows with the list (every event re-locates the whole list), so the
    # comparable figure is photon-steps per second
    sweep = None
    if rank == 0 and world == 1 and args.mode == "ranks" and args.other_mode and args.config != "cfg5":
        try:
            sweep = {"note": "the same %d photons as lists of about 1000 / 2000 / 5000 photons (the reference's range of photons per rank); one hydro "
                             "frame per step, 5 timed frames after 1; photon-steps/s is the comparable figure -- an event re-locates its whole list, "
                             "so events/s falls as the lists grow" % n, "runs": []}
            for per in (1000, 2000, 5000):
                lay = list_layout(per)
                ev_w, ps_w, it_w, dt_w = run_pooled(0, lay[0], 5, 1, layout=lay)
                sweep["runs"].append({"rank_photons": per, "lists": int(lay[0]), "ms_per_step": dt_w * 1e3 / 5, "photon_steps_per_s": ps_w / dt_w,
                                      "scatter_events_per_s": ev_w / dt_w, "passes_per_list": it_w / 5.0 / lay[0]})
            r_ = [x["photon_steps_per_s"] for x in sweep["runs"]]
            sweep["max_over_min"] = max(r_) / min(r_)
        except Exception as ex:
            sweep = {"error": "%s: %s" % (type(ex).__name__, ex)}

    other = None
    if rank == 0 and world == 1 and args.other_mode:
        if args.mode == "ranks":
            r = measure_list(500, 50, 200)
            other = {"mode": "list", "note": "one list of %d photons with one clock, one loop pass per step" % n, "steps": 500,
                     "ms_per_step": r["seconds"] * 1e3 / 500, "scatter_events_per_s": r["events"] / r["seconds"],
                     "photon_steps_per_s": r["photon_steps"] / r["seconds"], "roofline": r["roofline"],
                     "relocations_per_pass": r["relocations_per_pass"]}
        else:
            r = measure_ranks(5, 1, True)
            other = {"mode": "ranks", "rank_photons": args.rank_photons, "ranks": r["ranks"], "steps": 5,
                     "ms_per_step": r["seconds"] * 1e3 / 5, "scatter_events_per_s": r["events"] / r["seconds"],
                     "photon_steps_per_s": r["photon_steps"] / r["seconds"], "roofline": r["roofline"]}

    # the one-list-over-all-GPUs mode, timed briefly beside the main result.  It is the only part of this file with a
    # data-path collective; a watchdog makes sure that a stuck collective costs this extra, not the bench line.
    shared = None
    if args.mode != "shared-clock" and args.shared_clock_rounds > 0:
        import threading
        state = {"line": None}

        def give_up():
            if rank == 0 and state["line"] is not None:
                state["line"]["shared_clock"] = {"error": "no result within 120 s"}
                print(json.dumps(state["line"]), flush=True)
            os._exit(0)

        timer = threading.Timer(120.0, give_up)
        timer.daemon = True
    else:
        timer = None

    # what a drop-in caller pays per hydro frame around the loop: staging the frame (the lookup grid is built on the
    # device), the photons in as struct photon records, one frame of propagation, the photons out
    def plain_engine():
        return engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream, rng_stream=first_stream,
                             virtual_rank_photons=1000)

    def measure_pcie():
        """(a) the drop-in of INTEGRATION.md's first section: the caller's struct photon array and hydro columns in host memory, one
        context, one hydro frame = set_hydro + set_photons + propagate_frame + get_photons (buffers allocated once, as MCRaT's are).
        (b) the rank-pool driver of the host C (mcrat_host_run_ranks): the ranks' lists resident from device-side injection to the last
        frame; per frame the reader's buffers go in (ingest) and what the reference writes per frame comes out (checkpoint records,
        printPhotons' columns)."""
        aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
        out_buf = aos.copy()
        e = plain_engine()
        t = {"set_hydro": 0.0, "set_photons": 0.0, "propagate": 0.0, "get_photons": 0.0}
        reps, ev = 3, 0
        for k in range(reps + 1):
            t0 = time.perf_counter(); e.set_hydro(frame)
            t1 = time.perf_counter(); e.set_photons_aos(aos, num_null=0)
            t2 = time.perf_counter(); _, st = e.propagate_frame(0.0, remaining, SEED + 9000 + k)
            t3 = time.perf_counter(); e.get_photons_aos(out=out_buf)
            t4 = time.perf_counter()
            if k:                                             # the first round pays the allocations
                t["set_hydro"] += t1 - t0; t["set_photons"] += t2 - t1; t["propagate"] += t3 - t2; t["get_photons"] += t4 - t3
                ev += st.frame_scatt_cnt
        e.close()
        tot = sum(t.values())
        pcie = {"host_list": {"note": "per hydro frame through the C ABI with host-resident inputs and outputs: mcrat_hip_set_hydro + set_photons + "
                                      "propagate_frame (1000-photon virtual ranks) + get_photons, %d photons as struct photon records (176 MB each way)" % n,
                              "ms": {k: v * 1e3 / reps for k, v in t.items()}, "ms_per_frame": tot * 1e3 / reps,
                              "scatter_events_per_s": ev / tot}}
        if args.host_driver:
            try:
                pcie["rank_pool_driver"] = measure_host_driver()
            except Exception as ex:
                pcie["rank_pool_driver"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        return pcie

    def measure_host_driver():
        import ctypes as C
        import shutil
        import tempfile
        from mcrat_amd.host import binding as B
        host, h5 = B.host(), B.host_h5()
        side = 2.5e8 * (64 // args.nzc)
        raw = synth.flash_raw_blocks(side, args.nzc, 2 * args.nzc, args.nzc, 1e12 - args.nzc * side, seed=1)
        jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=3e50, theta_j=0.1)
        R, frames = 1024, 3
        res = {"note": "mcrat_host_run_ranks (host C): %d adopted ranks, each injecting 500-1000 photons on the device at frame 0 and scattering "
                       "through %d hydro frames of the cfg2 mesh read as a FLASH checkpoint (host buffers -> mcrat_hip_ingest_flash, once per frame "
                       "for all ranks); photons resident throughout.  ms per hydro frame, by what is written per rank and frame" % (R, frames),
               "ranks": R, "hydro_frames": frames}
        # (no_output_two_frames_per_launch: the same with mcrat_host_pool_config.stage_ctx -- frame F + 1 staged on a second context, two hydro frames per
        # launch, each frame's statistics from the captured lists; four frames = two launches)
        for label, chk, hdf in (("no_output", 0, 0), ("no_output_two_frames_per_launch", 0, 0), ("checkpoints", 1, 0), ("checkpoints_and_hdf5", 1, 1)):
            if hdf and h5 is None:
                continue
            staged = label.endswith("two_frames_per_launch")
            frames = 4 if staged else 3
            tmp = tempfile.mkdtemp(prefix="mcrat_bench_")
            try:
                pool = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0, device=local_rank, stream=stream)
                stage = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0, device=local_rank, stream=stream) if staged else None
                ranks = (B.HostRank * R)()
                for r, k in enumerate(ranks):
                    k.myid, k.angle_id, k.angle_procs = r, r, R
                    k.mc_dir = (tmp + "/").encode()
                    k.theta_jmin_thread, k.theta_jmax_thread, k.inj_radius, k.ph_weight_suggest = 0.0, 3.0 * np.pi / 180, 1e12, 1e50
                    k.framestart, k.frm2, k.rng_seed, k.rng_stream = 0, 0, SEED, 7000 + r
                pc = B.PoolConfig()
                pc.fps, pc.last_frm = float(frame["fps"]), frames - 1
                pc.r0_domain[0], pc.r0_domain[1] = frame["r0_domain"]
                pc.r1_domain[0], pc.r1_domain[1] = frame["r1_domain"]
                pc.spect, pc.min_photons, pc.max_photons = b"b", 500, 1000

                def reader(user, ctx, f, slab, pool=pool, stage=stage):
                    sl = slab.contents
                    (stage if (stage is not None and ctx == stage.ctx.value) else pool).ingest(raw, dict(r_inj=sl.r_inj, ph_inj_switch=sl.ph_inj_switch, min_r=sl.min_r, max_r=sl.max_r, min_theta=sl.min_theta,
                                          max_theta=sl.max_theta, fps=sl.fps, r0_domain=tuple(sl.r0_domain), r1_domain=tuple(sl.r1_domain),
                                          r2_domain=tuple(sl.r2_domain)), jet)
                    return 0
                pc.get_hydro = B.GET_HYDRO(reader)
                pc.write_checkpoints = chk
                if hdf:
                    pc.print_photons = C.cast(h5.mcrat_host_print_photon_arrays, C.c_void_p).value
                pc.comv_switch, pc.stokes_switch, pc.save_type = 1, 0, 0
                if staged:
                    pc.stage_ctx = stage.ctx.value
                t0 = time.perf_counter()
                rc = host.mcrat_host_run_ranks(pool.ctx, ranks, R, C.byref(pc))
                wall = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("mcrat_host_run_ranks: %d" % rc)
                photons = sum(k.num_photons for k in ranks)
                events = sum(k.frame_scatt_cnt_total for k in ranks)
                res[label] = {"photons": int(photons), "scatter_events": int(events), "wall_ms_total": wall * 1e3,
                              "ms_per_frame": {"propagate_and_statistics": pc.ms_propagate / frames, "output": pc.ms_output / frames,
                                               "hydro_reader_and_ingest": pc.ms_hydro / pc.hydro_frames_read},
                              "scatter_events_per_s_inclusive": events / ((pc.ms_propagate + pc.ms_output + pc.ms_hydro) * 1e-3)}
                if staged:
                    res[label]["hydro_frames"] = frames
                    res[label]["launches"] = int(pc.launches)
                    res[label]["two_frame_launches"] = int(pc.two_frame_launches)
                    stage.close()
                if chk or hdf:
                    # the writer thread (mcrat_hip_outbox_*): what it spent on the frames' files, how much of that the loop waited for, the rest
                    # was hidden behind the loop; and what the file system alone takes for as many files of the same size (same C, same box)
                    res[label]["ms_per_frame"]["writer_thread"] = pc.ms_output_writer / frames
                    res[label]["ms_per_frame"]["loop_blocked_on_writer"] = pc.ms_output_blocked / frames
                    res[label]["ms_per_frame"]["writer_hidden_behind_loop"] = max(pc.ms_output_writer - pc.ms_output_blocked, 0.0) / frames
                    ms_floor = C.c_double(0)
                    per_rank = 29 + 176 * int(pc.slots_per_rank or pc.max_photons)
                    if host.mcrat_host_output_floor((tmp + "/").encode(), R, per_rank, frames, 4, C.byref(ms_floor)) == 0:
                        res[label]["ms_per_frame"]["file_system_floor_checkpoints"] = ms_floor.value
                pool.close()
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        # a CONTINUE run's start (readCheckpoint's lists handed to the pool, mcrat.c:487): R lists of 1000 records from host memory, all at once
        # (mcrat_hip_pool_set_photons: one copy over PCIe, one launch) and list by list through the views (round 2's path)
        try:
            fr2, ph2, _ = synth.config2(n_photons=1000 * 64, nzc=args.nzc)
            recs = synth.photons_to_aos(ph2, engine.PHOTON_DTYPE)
            lists = [recs[(r % 64) * 1000:(r % 64 + 1) * 1000] for r in range(R)]
            pool = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0, device=local_rank, stream=stream)
            pool.pool_create(R, 1000)
            views = [pool.pool_rank(r, 7000 + r) for r in range(R)]
            pool.pool_set_photons(list(range(R)), lists)             # (allocations)
            pool.synchronize()
            t0 = time.perf_counter()
            pool.pool_set_photons(list(range(R)), lists)
            pool.synchronize()
            t_all = time.perf_counter() - t0
            t0 = time.perf_counter()
            for r in range(64):
                views[r].set_photons_aos(lists[r])
            pool.synchronize()
            t_each = (time.perf_counter() - t0) * R / 64
            res["restart_set_photons"] = {"ranks": R, "photons_per_rank": 1000, "ms_all_at_once": t_all * 1e3,
                                          "ms_list_by_list": t_each * 1e3, "note": "list by list: 64 lists timed, scaled to %d" % R}
            pool.close()
        except Exception as ex:
            res["restart_set_photons"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        return res

    pcie = None
    if rank == 0 and world == 1 and args.mode == "ranks" and args.other_mode and args.config == "cfg2":
        try:                                    # an extra must never cost the bench line
            pcie = measure_pcie()
        except Exception as ex:
            pcie = {"error": "%s: %s" % (type(ex).__name__, ex)}

    # the step in front of the loop: a FLASH checkpoint's datasets (host buffers, as H5Dread leaves them)
    # -> expansion, slab selection, structured-jet overwrite, staged frame with its lookup grid -> device-side injection
    def measure_ingest():
        side = 2.5e8 * (64 // args.nzc)
        raw = synth.flash_raw_blocks(side, args.nzc, 2 * args.nzc, args.nzc, 1e12 - args.nzc * side, seed=1)
        slab = dict(r_inj=1e12, ph_inj_switch=1, min_r=0.0, max_r=0.0, min_theta=0.0, max_theta=0.0, fps=float(frame["fps"]),
                    r0_domain=frame["r0_domain"], r1_domain=frame["r1_domain"], r2_domain=(0.0, 0.0))
        jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=3e50, theta_j=0.1)
        e = plain_engine()
        t_in, t_inj, reps = 0.0, 0.0, 3
        for k in range(reps + 1):
            t0 = time.perf_counter(); m, ef, cells = e.ingest(raw, slab, jet)
            t1 = time.perf_counter(); nph, _ = e.inject_photons(1e12, 1e50, n // 2, n, "b", 0.0, 3.0 * np.pi / 180, float(frame["fps"]), SEED + k)
            t2 = time.perf_counter()
            if k:
                t_in += t1 - t0; t_inj += t2 - t1
        e.close()
        ingest = {"note": "mcrat_hip_ingest_flash on a synthetic FLASH checkpoint of the cfg2 mesh (host buffers in pageable memory: H2D, "
                          "leaf-block expansion, selection of the injection frame's cells, structured-jet overwrite, per-cell records, "
                          "cell-lookup grid), then mcrat_hip_inject_photons on the staged frame",
                  "cells_read": int(cells), "cells_selected": int(m), "elem_factor": int(ef), "photons_injected": int(nph),
                  "ms_ingest": t_in * 1e3 / reps, "ms_inject": t_inj * 1e3 / reps, "cells_per_s": cells * reps / t_in}
        if not args.no_cpu_baseline:
            from oracle import oracle_py
            ocfg = oracle_py.make_config(cfg["dimensions"], cfg["geometry"], 0)
            t0 = time.perf_counter()
            ref, _ = oracle_py.hydro_ingest(ocfg, raw, slab, oracle_py.outflow(3, lumi=3e50, theta_j=0.1))
            ingest["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
            ingest["cpu_oracle_note"] = "oracle/oracle_ingest.c on one host core: selection and overwrite only (no lookup grid, no injection)"
            ingest["cpu_oracle_cells_selected"] = int(ref["num_elements"])
        return ingest

    ingest = None
    if rank == 0 and world == 1 and args.mode == "ranks" and args.other_mode and args.config == "cfg2":
        try:                                    # an extra must never cost the bench line
            ingest = measure_ingest()
        except Exception as ex:
            ingest = {"error": "%s: %s" % (type(ex).__name__, ex)}

    def measure_hot():
        """the same jet ten times closer to the engine: every cell above 1e7 K, where the reference's Maxwell-Juttner sampler accepts one
        attempt in ~150 and the event walk's wavefront tries 64 at a time; 300 passes per list"""
        hframe, hph, hcfg = synth.config2(n_photons=args.photons, seed=SEED + rank, nzc=args.nzc, stokes=args.stokes, lumi=1e54, r_inj=1e11,
                                          block_side=2.5e7)
        e = engine.Engine(hcfg["dimensions"], hcfg["geometry"], hcfg["stokes"], device=local_rank, stream=stream, rng_stream=first_stream,
                          virtual_rank_photons=1000)
        e.set_hydro(hframe)
        e.set_photons(hph)
        e.snapshot_photons()
        best = None
        for _ in range(3):
            e.restore_photons()
            e.begin_frame(SEED, 0.0, 1.0 / hframe["fps"])
            e.run(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = e.run(300)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, int(st.frame_scatt_cnt), int(st.iterations))
        e.close()
        return {"workload": "cfg2 jet at r_inj = 1e11 cm, L = 1e54 erg/s: T' = %.1e .. %.1e K; %d lists x %d photons, 300 passes each"
                            % (float(hframe["temp"].min()), float(hframe["temp"].max()), (args.photons + 999) // 1000, 1000),
                "scatter_events_per_s": best[1] / best[0], "ms": best[0] * 1e3, "scatter_events": best[1], "loop_passes": best[2]}

    hot = None
    if rank == 0 and world == 1 and args.mode == "ranks" and args.other_mode and args.config == "cfg2":
        try:
            hot = measure_hot()
        except Exception as ex:
            hot = {"error": "%s: %s" % (type(ex).__name__, ex)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            if args.mode == "ranks":
                cpu = cpu_baseline_ranks(frame, ph, cfg, args.rank_photons, host_cores())
            else:
                cpu = cpu_baseline_list(frame, ph, cfg, min(1024, n), 300)
        except Exception as ex:                 # report, do not lose the line
            cpu = {"error": "%s: %s" % (type(ex).__name__, ex)}
    cpu_opt = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "ranks" and args.other_mode:
        try:
            cpu_opt = cpu_optimised_ranks(frame, ph, cfg, args.rank_photons, host_cores())
        except Exception as ex:
            cpu_opt = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        if args.mode == "ranks":
            pools = 1 if args.launch_shape == "queue" else max(1, min(int(args.pools) if args.pools > 0 else 3, n_lists))
            shape = ("%d adopted ranks with lists of %d-%d photons (independent lists, own clock and RNG stream each: the reference's MPI "
                     "ranks; one workgroup per list) in %s; step = one hydro frame (1/fps = %.2f s) for all lists, every pool restarted from "
                     "its resident snapshot"
                     % (main_res.get("ranks", 0), int(lens.min()), int(lens.max()),
                        ("one rank pool and ONE call for all %d timed frames (mcrat_hip_pool_run_frames; roofline.launches says how many launches it "
                         "took: one -- the frame queue, a workgroup per (frame, list) item, a list that is through frame f starts f + 1 while others are "
                         "still in f, as the reference's ranks are asynchronous across hydro frames, mcrat.c:457-479,566-934 -- where the launch form has a "
                         "queue build (256-thread lists in LDS), else one per frame)" % steps) if args.launch_shape == "queue" else
                        "one rank pool, all lists in one launch" if pools == 1 else
                        "%d rank pools on %d HIP streams with a host thread each (the ranks are asynchronous in the reference too: a pool whose "
                        "last lists are finishing no longer leaves the device idle), one launch per pool and frame" % (pools, pools), remaining))
        elif args.mode == "list":
            shape = "one list, one clock; step = one loop pass over all photons"
        else:
            shape = ("ONE list of %d photons with one clock, its slots spread over %d GPU(s); step = one round: own slots "
                     "stepped, 736-B proposals all-gathered (RCCL), merged candidates walked on every GPU" % (n * world, world))
        out = {
            "metric": "photon-scatter-events/sec at 1e6 photons, 2D FLASH jet" if args.config == "cfg2" else
                      "photon-scatter-events/sec at 1e7 photons, 2D PLUTO spherical jet, Stokes on",
            "value": events_all / t_max,
            "unit": "scatter-events/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": t_max * 1e3 / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json %s (%d cells, Lundman structured "
                                   "jet), %d photons per GPU, Compton+KN, STOKES %s; %s"
                                   % ("configs[1]: 2D FLASH-like cylindrical GRB-jet frame" if args.config == "cfg2" else
                                      "configs[2]: 2D PLUTO-like spherical (log r, theta) jet frame", frame["num_elements"], n,
                                      "on" if args.stokes else "off", shape),
                       "mode": args.mode, "photons_per_gpu": n, "cells": int(frame["num_elements"]),
                       "rank_photons": args.rank_photons if args.mode == "ranks" else None,
                       "parallelism": ("one list, slots sharded x%d, all-gather per round" % world) if args.mode == "shared-clock"
                                      else "independent photon shards x%d" % world},
            "photon_steps_per_s": steps_all / t_max,
            "scatter_events": events_all,
            "loop_passes": main_res["passes"],
            "roofline": main_res["roofline"],
            "strong": strong,
            "rank_photons_sweep": sweep,
            "fast_mode": fast,
            "other_mode": other,
            "pcie_inclusive": pcie,
            "ingest": ingest,
            "hot_frame": hot,
            "cpu_baseline": cpu,
            "cpu_optimised": cpu_opt,
        }
    else:
        out = None
    if timer is not None:
        state["line"] = out
        timer.start()
        try:
            r = measure_shared(args.shared_clock_rounds, 20)
            tt = torch.tensor([r["seconds"]], dtype=torch.float64, device=red_dev)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sec = float(tt.item())
            shared = {"mode": "shared-clock", "note": "ONE list of %d photons with one clock, slots spread over %d GPU(s); per "
                      "round one all-gather of 736 B per GPU (RCCL) and a replicated photonEvent walk" % (n * world, world),
                      "photons_total": n * world, "rounds": r["rounds"], "loop_passes": r["passes"], "midpass_rounds": r["midpass_rounds"],
                      "ms_per_round": sec * 1e3 / r["rounds"], "scatter_events_per_s": r["events"] / sec,
                      "photon_steps_per_s": float(n) * world * r["passes"] / sec}
        except BaseException as ex:          # the bench line matters more than this extra
            shared = {"error": "%s: %s" % (type(ex).__name__, ex)}
            if out is not None:
                out["shared_clock"] = shared
                print(json.dumps(out), flush=True)
            os._exit(0)
        timer.cancel()
        if out is not None:
            out["shared_clock"] = shared
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
