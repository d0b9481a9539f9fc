"""mcrat_host_run_ranks at the size one GPU of BASELINE.json configs[3] would hold (1e8 photons over 8 GPUs: ~1.25e7 photons, ~12 800 adopted ranks
per GPU): R ranks injecting 500-1000 photons each on the device, FRAMES hydro frames of the cfg2 mesh read as a FLASH checkpoint, photons resident,
no output files; EXACT then FAST.  Prints the per-frame costs and the device memory in use before and after (leaks would show).
    python tools/driver_stress.py [ranks] [frames]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mcrat_amd import engine, synth  # noqa: E402
from mcrat_amd.host import binding as B  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 20
host = B.host()
raw = synth.flash_raw_blocks(2.5e8, 64, 128, 64, 1e12 - 64 * 2.5e8, seed=1)
jet = engine.Engine.outflow(engine.STRUCTURED_SPHERICAL_OUTFLOW, lumi=3e50, theta_j=0.1)


def used_gb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 30


for mode in ("exact", "fast"):
    base = used_gb()
    pool = engine.Engine(synth.TWO, synth.CYLINDRICAL, 0)
    ranks = (B.HostRank * R)()
    for r, k in enumerate(ranks):
        k.myid, k.angle_id, k.angle_procs = r, r, R
        k.mc_dir = b"/tmp/"
        k.theta_jmin_thread, k.theta_jmax_thread, k.inj_radius, k.ph_weight_suggest = 0.0, 3.0 * np.pi / 180, 1e12, 1e50
        k.framestart, k.frm2, k.rng_seed, k.rng_stream = 0, 0, 0x4D435261, 7000 + r
    pc = B.PoolConfig()
    pc.fps, pc.last_frm = 5.0, FRAMES - 1
    pc.r0_domain[0], pc.r0_domain[1] = 0.0, 5e12
    pc.r1_domain[0], pc.r1_domain[1] = 0.0, 2.5e13
    pc.spect, pc.min_photons, pc.max_photons = b"b", 500, 1000
    frame_ms = []

    def reader(user, ctx, f, slab, pool=pool):
        sl = slab.contents
        t0 = time.perf_counter()
        pool.ingest(raw, dict(r_inj=sl.r_inj, ph_inj_switch=sl.ph_inj_switch, min_r=sl.min_r, max_r=sl.max_r, min_theta=sl.min_theta, max_theta=sl.max_theta,
                              fps=sl.fps, r0_domain=tuple(sl.r0_domain), r1_domain=tuple(sl.r1_domain), r2_domain=tuple(sl.r2_domain)), jet)
        frame_ms.append((time.perf_counter(), (time.perf_counter() - t0) * 1e3))
        return 0
    pc.get_hydro = B.GET_HYDRO(reader)
    pc.write_checkpoints = 0
    pc.comv_switch, pc.stokes_switch, pc.save_type = 1, 0, 0
    pc.mode, pc.fast_windows = (1, 8) if mode == "fast" else (0, 0)
    t0 = time.perf_counter()
    rc = host.mcrat_host_run_ranks(pool.ctx, ranks, R, C.byref(pc))
    wall = time.perf_counter() - t0
    assert rc == 0, rc
    photons = sum(k.num_photons for k in ranks)
    events = sum(k.frame_scatt_cnt_total for k in ranks)
    stamps = [t for t, _ in frame_ms]
    per_frame = np.diff(stamps[1:]) * 1e3                      # wall time from one scatter frame's read to the next (the first read is the injection's)
    print("%-5s  %d ranks, %d photons, %d hydro frames: %.1f s in all (injection of all ranks included); %d scatterings" % (mode, R, photons, FRAMES, wall, events))
    print("       per hydro frame: propagate + statistics %.2f ms, reader + ingest %.2f ms; frame-to-frame wall: first %.2f ms, median %.2f ms, last %.2f ms"
          % (pc.ms_propagate / FRAMES, pc.ms_hydro / pc.hydro_frames_read, per_frame[0], float(np.median(per_frame)), per_frame[-1]))
    print("       device memory in use: %.2f GB with the pool, %.2f GB before" % (used_gb(), base), flush=True)
    pool.close()
    torch.cuda.synchronize()
    print("       after closing the pool: %.2f GB" % used_gb(), flush=True)
