#!/usr/bin/env python3
"""bench.py -- the photon loop on the BASELINE.json workload.

    python bench.py --gpus 1 --steps 2000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one pass of MCRaT's `while (remaining_time > 0)` body (Src/mcrat.c:761-851) over the
whole photon list: re-locate every photon, draw every free path, pick the earliest, advance all, scatter
one -- i.e. one launch of the step kernel + one of the event kernel.  Workload: configs[1] of
BASELINE.json (synthetic 2-D FLASH-like cylindrical GRB-jet frame, 1 048 576 cells, 1e6 photons),
photons and hydro frame resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `value` = scatter events per second over all ranks; the same line carries
photon-steps/s (slots x iterations / s, the quantity the HBM roofline is priced on), the roofline object
of the step kernel and the CPU baseline (oracle = faithful restatement of the reference's algorithm,
timed on this box's host cores on a bounded sample).
"""
import argparse
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


def committed_traffic():
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (profiles/*_step_kernel_pmc.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as the gfx950 note of
    MI355X_MICROARCH.md prescribes).  bench.py itself does not run under the profiler, so this is the latest
    committed measurement of the same kernel on the same workload, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_step_kernel_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return d.get("traffic_bytes_per_launch"), os.path.basename(files[-1])


def cpu_baseline(frame, ph, cfg, seed, n_sample, iters):
    """The oracle (oracle/, plain-C restatement of the reference: 176-B AoS photons, full redraw and full
    qsort_r argsort per event, linear cell search, one thread) on a bounded sample of the same workload."""
    from mcrat_amd import synth
    from oracle import oracle_py as O
    sub = {k: (v[:n_sample].copy() if isinstance(v, np.ndarray) else v) for k, v in ph.items()}
    H = O.OracleHydro(frame)
    P = O.OraclePhotons(synth.photons_to_aos(sub, O.PHOTON_DTYPE))
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    rem = 1.0 / frame["fps"]
    t0 = time.perf_counter()
    st1, tn, rem, sw = O.photon_loop(c, P, H, seed=seed, time_now=0.0, remaining_time=rem, max_iterations=1)
    t1 = time.perf_counter()
    # the jet is optically thin: with a few thousand photons a frame holds only a handful of events, so the
    # sample keeps iterating on the frozen frame (remaining_time re-armed) until `iters` passes are timed
    done, scatt, base = 0, 0, 1
    while done < iters:
        if rem <= 0:
            rem = 1.0 / frame["fps"]
        st, tn, rem, sw = O.photon_loop(c, P, H, seed=seed, time_now=tn, remaining_time=rem,
                                        max_iterations=iters - done, iteration_base=base, find_switch=sw)
        done += st.iterations
        base += st.iterations
        scatt += st.frame_scatt_cnt
        if st.iterations == 0:
            break
    t2 = time.perf_counter()
    dt = t2 - t1
    return {
        "value": scatt / dt,
        "unit": "scatter-events/s",
        "cores": 1,
        "kind": "port",
        "photon_steps_per_s": n_sample * done / dt,
        "first_pass_s": t1 - t0,
        "sample": ("oracle/ (faithful C restatement; reference itself needs GSL and cannot be built here), 1 thread: "
                   "first %d photons of the same photon set on the same %d-cell frame, %d loop iterations timed after "
                   "the forced O(N*M) re-location pass (that pass alone took first_pass_s); per-event cost grows as "
                   "N log N, compare photon_steps_per_s" % (n_sample, frame["num_elements"], done)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--photons", type=int, default=1_000_000, help="photon slots per GPU")
    ap.add_argument("--nzc", type=int, default=64, help="mesh scale: 64 -> 1 048 576 cells")
    ap.add_argument("--stokes", type=int, default=0)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--profile-steps", type=int, default=300)
    ap.add_argument("--cpu-photons", type=int, default=1024)
    ap.add_argument("--cpu-steps", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- workload: every rank owns an independent photon set on a replica of the frame (weak scaling;
    # the reference's ranks own disjoint photons and never talk during the loop)
    seed = 0x4D435261
    frame, ph, cfg = synth.config2(n_photons=args.photons, seed=seed + rank, nzc=args.nzc, stokes=args.stokes)
    n = int(ph["p0"].size)
    per_sync = max(50, min(500, args.steps))
    stream = torch.cuda.current_stream().cuda_stream
    eng = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream,
                        rng_stream=rank, iterations_per_sync=per_sync, use_graph=bool(args.graph))
    eng.set_hydro(frame)
    eng.set_photons(ph)
    remaining = 1.0 / frame["fps"]
    eng.begin_frame(seed, 0.0, remaining)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    w = eng.run(args.warmup) if args.warmup > 0 else None      # includes the forced re-location pass
    it0 = w.iterations if w else 0
    sc0 = w.frame_scatt_cnt if w else 0
    sync()
    t0 = time.perf_counter()
    st = eng.run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    steps_done = st.iterations - it0
    scatt = st.frame_scatt_cnt - sc0
    if steps_done != args.steps:
        raise SystemExit("the frame ended after %d of %d timed steps; lower --steps" % (steps_done, args.steps))

    t_max, scatt_all, slots_all = dt, scatt, n
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([float(scatt), float(n)], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t_max, scatt_all, slots_all = float(t.item()), float(c[0].item()), float(c[1].item())

    # ---- roofline of the step kernel: its launches bracketed by HIP events on the engine's stream, in a
    # separate pass that continues the same frame (event records perturb the timed region, so they are not in it)
    roof = None
    if rank == 0 and args.profile_steps > 0:
        prof = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=local_rank, stream=stream,
                             rng_stream=rank, iterations_per_sync=min(per_sync, args.profile_steps), profile=True)
        prof.set_hydro(frame)
        prof.set_photons(eng.get_photons())
        prof.begin_frame(seed + 1, 0.0, remaining)
        prof.step_locate_sample(0)       # the photons are already located: skip the forced re-location pass
        p0 = prof.run(20)                # warm caches; not counted
        ps = prof.run(args.profile_steps)
        launches = ps.step_kernel_launches - p0.step_kernel_launches
        avg_ms = (ps.step_kernel_ms - p0.step_kernel_ms) / max(1, launches)
        ev_ms = (ps.event_kernel_ms - p0.event_kernel_ms) / max(1, launches)
        achieved = ALGORITHMIC_BYTES_PER_PHOTON_STEP * n / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic() if (n == 1_000_000 and args.nzc == 64) else (None, None)
        roof = {
            "kernel": "step_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": ALGORITHMIC_BYTES_PER_PHOTON_STEP * n,
            "avg_launch_ms": avg_ms, "launches": int(launches), "event_kernel_avg_ms": ev_ms,
        }
        prof.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(frame, ph, cfg, seed, min(args.cpu_photons, n), args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "photon-scatter-events/sec at 1e6 photons, 2D FLASH jet",
            "value": scatt_all / t_max,
            "unit": "scatter-events/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 2D FLASH-like cylindrical GRB-jet frame (%d cells, "
                                   "Lundman structured jet), %d photons per GPU, Compton+KN, STOKES %s, exact "
                                   "event-driven loop" % (frame["num_elements"], n, "on" if args.stokes else "off"),
                       "photons_per_gpu": n, "cells": int(frame["num_elements"]), "parallelism": "independent photon shards x%d" % world,
                       "graph": bool(args.graph)},
            "photon_steps_per_s": slots_all * args.steps / t_max,
            "scatter_events": scatt_all,
            "relocations_per_step": (st.num_photons_find_new_element - (w.num_photons_find_new_element if w else 0)) / args.steps,
            "kn_rejections": st.kn_rejections, "rescans": st.rescans,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
