"""mcrat_host_run_ranks (host C): one process adopts R of the reference's MPI ranks -- main()'s rank/angle split, injection
frames and scatter-frame loops (mcrat.c:116-164,457-483,609-924) on top of the rank pool.  Eight ranks in two angle bins with
Poisson-sized lists that join at their own injection frames; every rank's files must be what that rank alone would write:
  - the checkpoint files byte for byte those of a single-list context doing the rank's steps by hand,
  - the HDF5 datasets of mc_proc_<rank>.h5 equal to that context's printPhotons,
  - the photons within 1e-9 of the oracle running the rank's frames on its own."""
import ctypes as C
import os

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu

DOM = dict(r0_domain=(1e11, 4e12), r1_domain=(0.0, 0.6), r2_domain=(0.0, 0.0))
FPS, LAST = 5.0, 4


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _mcpar(B):
    p = B.McPar()
    p.fps, p.last_frame = FPS, LAST
    for k, v in DOM.items():
        getattr(p, k)[0], getattr(p, k)[1] = v
    p.theta_jmin, p.theta_j, p.n_theta_j = 0.0, 6.0, 2
    keep = [(C.c_int * 2)(0, 1), (C.c_int * 2)(4, 3), (C.c_double * 2)(1e12, 1.2e12)]      # frm0; frm2 = frm0 + number of frames; inj_radius
    p.frm0, p.frm2, p.inj_radius = keep[0], keep[1], keep[2]
    p.spect, p.min_photons, p.max_photons, p.restart = b"b", 300, 900, b"i"
    return p, keep


def _slab_dict(s):
    return dict(r_inj=s.r_inj, ph_inj_switch=s.ph_inj_switch, min_r=s.min_r, max_r=s.max_r, min_theta=s.min_theta, max_theta=s.max_theta, fps=s.fps,
                r0_domain=tuple(s.r0_domain), r1_domain=tuple(s.r1_domain), r2_domain=tuple(s.r2_domain))


@pytest.mark.parametrize("mode", ["exact", "fast", "exact-sync"])
def test_eight_ranks_through_the_host_c_equal_eight_ranks_on_their_own(hip, oracle, tmp_path, mode):
    """The files are written by the writer thread while the next frame propagates (mcrat_hip_outbox_*, the default); "exact-sync": sync_output = 1,
    the reference's order -- the same bytes either way.  mode "fast": mcrat_host_pool_config.mode = MCRAT_HIP_MODE_FAST -- every rank's frame through mcrat_hip_pool_propagate_frames_fast with the
    rank's own seed and stream; its files are then byte for byte those of a context doing the rank's steps by hand in FAST mode (there is no
    oracle of the FAST sequence: tests/test_gpu_fast_mode.py holds it against the exact mode in distribution)"""
    fast = mode == "fast"
    sync_output = mode.endswith("-sync")
    from mcrat_amd.host import binding as B
    host, h5 = B.host(), B.host_h5()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    jet = hip.Engine.outflow(3, lumi=2e53, theta_j=0.1)
    par, keep = _mcpar(B)
    R, base_seed = 8, 20261004
    ranks = (B.HostRank * R)()
    base = str(tmp_path / "pool") + "/"
    assert host.mcrat_host_split_ranks(C.byref(par), R, 0, R, base.encode(), 1e50, base_seed, ranks) == 0
    # the split of mcrat.c:139-164,457-483: two bins of four ranks, one block of injection frames each
    assert [(k.angle_id, k.angle_procs) for k in ranks] == [(a, 4) for a in range(4)] * 2
    assert [(k.framestart, k.frm2) for k in ranks] == [(0, 0), (1, 1), (2, 2), (3, 4), (1, 1), (2, 2), (3, 3), (4, 3)]   # the last one has nothing to inject
    assert ranks[0].mc_dir.decode() == base + "0.0-3.0/" and ranks[7].mc_dir.decode() == base + "3.0-6.0/"
    assert ranks[5].inj_radius == 1.2e12 and ranks[5].theta_jmin_thread == pytest.approx(3 * np.pi / 180)
    logs = []
    for k in ranks:
        os.makedirs(k.mc_dir.decode(), exist_ok=True)
        k.fPtr = libc.fopen((k.mc_dir.decode() + "mc_output_%d.log" % k.angle_id).encode(), b"a")
        logs.append(k.fPtr)

    pool = hip.Engine(synth.TWO, synth.SPHERICAL, 1)
    slabs = []

    def reader(user, ctx, frame, slab):
        s = slab.contents
        slabs.append((frame, _slab_dict(s)))
        assert ctx == pool.ctx.value
        pool.ingest(raw, _slab_dict(s), jet)
        return 0
    cfg = B.PoolConfig()
    cfg.fps, cfg.last_frm = FPS, LAST
    for k, v in DOM.items():
        getattr(cfg, k)[0], getattr(cfg, k)[1] = v
    cfg.spect, cfg.min_photons, cfg.max_photons, cfg.slots_per_rank = b"b", 300, 900, 0
    cfg.get_hydro = B.GET_HYDRO(reader)
    cfg.write_checkpoints = 1
    if h5 is not None:
        cfg.print_photons = C.cast(h5.mcrat_host_print_photon_arrays, C.c_void_p).value
    cfg.comv_switch, cfg.stokes_switch, cfg.save_type = 1, 1, 1
    cfg.mode, cfg.fast_windows = (hip.MODE_FAST, 4) if fast else (hip.MODE_EXACT, 0)
    cfg.sync_output, cfg.output_threads = int(sync_output), 3
    assert host.mcrat_host_run_ranks(pool.ctx, ranks, R, C.byref(cfg)) == 0
    assert (cfg.ms_output_writer > 0) == (not sync_output) and cfg.ms_output_blocked <= cfg.ms_output + 1e-9
    for f in logs:
        libc.fclose(f)
    # every hydro frame was read once per pass over the frames for all ranks in it (+ the injection reads): not once per rank
    frames_read = [f for f, s in slabs if s["ph_inj_switch"] == 0]
    assert frames_read == [0, 1, 2, 3, 4, 4] and cfg.launches == 6
    inj_reads = [(f, s["r_inj"]) for f, s in slabs if s["ph_inj_switch"] == 1]
    assert inj_reads == [(0, 1e12), (1, 1e12), (1, 1.2e12), (2, 1e12), (2, 1.2e12), (3, 1e12), (3, 1.2e12), (4, 1e12)]
    sizes = [k.num_photons for k in ranks]
    assert all(300 <= n <= 900 for n in sizes[:7]) and len(set(sizes[:7])) > 3 and sizes[7] == 0     # Poisson-sized lists (mclib.c:87-136)

    # ---- every rank on its own, by hand, with the frames the pool saw
    def frame_slab(F, inj, r_inj=None):
        for f, s in slabs:
            if f == F and s["ph_inj_switch"] == inj and (r_inj is None or s["r_inj"] == r_inj):
                return s
        raise KeyError((F, inj))
    cfg_o = oracle.make_config(synth.TWO, synth.SPHERICAL, 1)
    total_scatt = 0
    for r, k in enumerate(ranks):
        d = k.mc_dir.decode()
        own = str(tmp_path / ("own_%d" % r)) + "/"
        os.makedirs(own)
        if k.framestart > k.frm2:
            assert not os.path.exists(d + "mc_proc_%d.h5" % k.angle_id)
            continue
        e = hip.Engine(synth.TWO, synth.SPHERICAL, 1, rng_stream=k.rng_stream, virtual_rank_photons=900)
        draws, file_frames = 0, {}
        for frame in range(k.framestart, k.frm2 + 1):
            if frame > LAST:
                continue
            e.ingest(raw, frame_slab(frame, 1, k.inj_radius), jet)
            n, w = e.inject_photons(k.inj_radius, 1e50, 300, 900, "b", k.theta_jmin_thread, k.theta_jmax_thread, FPS, B.rank_seed(base_seed, draws))
            draws += 1
            P = oracle.OraclePhotons(e.get_photons_aos())
            t = frame / FPS
            for F in range(frame, LAST + 1):
                sl = frame_slab(F, 0) if not (frame == 4 and F == 4 and r == 3) else [s for f, s in slabs if f == 4 and s["ph_inj_switch"] == 0][1]
                e.ingest(raw, sl, jet)
                seed = B.rank_seed(base_seed, draws)
                draws += 1
                rem = (F + 1) / FPS - t
                if fast:
                    t_new, st = e.propagate_frame_fast(t, rem, seed, 4)
                else:
                    t_new, st = e.propagate_frame(t, rem, seed)
                    ref_frame, _ = oracle.hydro_ingest(cfg_o, raw, sl, oracle.outflow(3, lumi=2e53, theta_j=0.1))
                    H = oracle.OracleHydro(dict(ref_frame, **DOM, fps=FPS))
                    ost, otime, _, _ = oracle.photon_loop(cfg_o, P, H, seed=seed, time_now=t, remaining_time=rem, stream=k.rng_stream)
                    assert (st.iterations, st.frame_scatt_cnt) == (ost.iterations, ost.frame_scatt_cnt)
                total_scatt += st.frame_scatt_cnt
                t = t_new
                assert host.mcrat_host_save_checkpoint(own.encode(), frame, k.frm2, F, t, e.ctx, None, n, LAST, k.angle_id, k.angle_procs, 0) == 0
                if h5 is not None:
                    assert h5.mcrat_host_print_photons(e.ctx, F, own.encode(), k.angle_id, 1, 1, 1, None) == 0
                file_frames.setdefault(F, 0)
                file_frames[F] += int((e.get_photons_aos()["weight"] != 0).sum())
            last_list = (e.get_photons_aos(), P.aos, t)
        # the checkpoint the pool left for this rank: the closing one of mcrat.c:924 (header only); the one before it is kept as _old
        final = open(d + "mc_chkpt_%d.dat" % k.angle_id, "rb").read()
        assert len(final) == 4 + 1 + 4 + 4 + 4 + 8 + 4
        assert host.mcrat_host_save_checkpoint(own.encode(), k.frm2 + 1, k.frm2, LAST + 1, last_list[2], None, None, 0, LAST, k.angle_id, k.angle_procs, 0) == 0
        for name in ("mc_chkpt_%d.dat" % k.angle_id, "mc_chkpt_%d.dat_old" % k.angle_id):
            assert open(d + name, "rb").read() == open(own + name, "rb").read(), (r, name)
        # ... whose records are the rank's photons after its last frame, within 1e-9 of the oracle's.  (A rank whose last injection frame
        # is the last hydro frame ends with frame == scatt_frame in the closing call, which removes the file instead of keeping it,
        # mcrat_io.c:900-903: its _old is an earlier frame's -- equal to the by-hand one above, not compared with the oracle here.)
        if k.frm2 != LAST and not fast:
            old = open(d + "mc_chkpt_%d.dat_old" % k.angle_id, "rb").read()
            rec = np.frombuffer(old[4 + 1 + 4 + 4:], dtype=hip.PHOTON_DTYPE)      # the 'i' layout of the last scatter frame (mcrat_io.c:974-998)
            assert old[4:5] == b"i" and len(rec) == len(last_list[0])
            assert np.array_equal(rec["num_scatt"], last_list[1]["num_scatt"])
            for col in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "s1", "s2"):
                scale = np.maximum(np.abs(last_list[1][col]), {"p": 0.0, "r": 1e9, "s": 1.0}[col[0]])
                if col[0] == "p":
                    scale = np.abs(last_list[1]["p0"])
                assert np.all(np.abs(rec[col] - last_list[1][col]) <= 1e-9 * scale), (r, col)
        # mc_proc_<rank>.h5: group per scatter frame; a rank with two injection frames appends to the groups it has already written (mcrat_io.c:402-424)
        if h5 is not None:
            for F, count in file_frames.items():
                for name, is_char in (("P0", 0), ("R2", 0), ("COMV_P3", 0), ("S1", 0), ("NS", 0), ("PW", 0), ("PT", 1)):
                    got = []
                    for path in (d, own):
                        n = C.c_int()
                        buf = np.empty(count, dtype="S1" if is_char else np.float64)
                        assert h5.mcrat_host_h5_read((path + "mc_proc_%d.h5" % k.angle_id).encode(), str(F).encode(), name.encode(), is_char,
                                                     buf.ctypes.data, count, C.byref(n)) == 0
                        assert n.value == count, (r, F, name, n.value, count)
                        got.append(buf)
                    assert np.array_equal(got[0], got[1]), (r, F, name)
        log = open(d + "mc_output_%d.log" % k.angle_id).read()
        assert log.count("Injecting photons") == min(k.frm2, LAST) - k.framestart + 1
        assert log.count("The number of scatterings in this frame is: ") == sum(LAST - f + 1 for f in range(k.framestart, min(k.frm2, LAST) + 1))
        e.close()
    assert total_scatt > 500
    assert sum(k.frame_scatt_cnt_total for k in ranks) == total_scatt
    pool.close()


def test_restarted_rank_continues_from_its_checkpoint(hip, tmp_path):
    """a CONTINUE run: the rank's list comes from mcrat_host_read_checkpoint and picks up at scatt_framestart with the saved clock"""
    from mcrat_amd.host import binding as B
    host = B.host()
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    jet = hip.Engine.outflow(3, lumi=2e53, theta_j=0.1)

    def run(ranks, n, max_frames, dirs):
        pool = hip.Engine(synth.TWO, synth.SPHERICAL, 1)
        cfg = B.PoolConfig()
        cfg.fps, cfg.last_frm = FPS, 3
        for k, v in DOM.items():
            getattr(cfg, k)[0], getattr(cfg, k)[1] = v
        cfg.spect, cfg.min_photons, cfg.max_photons = b"b", 300, 900
        cb = B.GET_HYDRO(lambda user, ctx, frame, slab: (pool.ingest(raw, _slab_dict(slab.contents), jet), 0)[1])
        cfg.get_hydro = cb
        cfg.write_checkpoints, cfg.max_frames = 1, max_frames
        for k, d in zip(ranks, dirs):
            os.makedirs(d, exist_ok=True)
            k.mc_dir = d.encode()
        assert host.mcrat_host_run_ranks(pool.ctx, ranks, n, C.byref(cfg)) == 0
        out = [pool.pool_rank(r, ranks[r].rng_stream).get_photons_aos() if ranks[r].num_photons else None for r in range(n)]
        pool.close()
        return out

    def fresh():
        ranks = (B.HostRank * 2)()
        for r, k in enumerate(ranks):
            k.myid, k.angle_id, k.angle_procs = r, r, 2
            k.theta_jmin_thread, k.theta_jmax_thread, k.inj_radius, k.ph_weight_suggest = 0.0, 0.05, 1e12, 1e50
            k.framestart, k.frm2, k.rng_seed, k.rng_stream = r, r, 99, r
        return ranks
    # straight through frames 0..3
    a = fresh()
    whole = run(a, 2, 0, [str(tmp_path / "a0") + "/", str(tmp_path / "a1") + "/"])
    # stopped after two hydro frames, then continued from the checkpoints
    b = fresh()
    d = [str(tmp_path / "b0") + "/", str(tmp_path / "b1") + "/"]
    run(b, 2, 2, d)
    c = fresh()
    lists = []
    for r, k in enumerate(c):
        l = hip.PhotonList()
        frm2, fs, sfs, ang = C.c_int(k.frm2), C.c_int(k.framestart), C.c_int(), C.c_int()
        restart, t = C.c_char(b"c"), C.c_double()
        assert host.mcrat_host_read_checkpoint(d[r].encode(), C.byref(l), C.byref(frm2), C.byref(fs), C.byref(sfs), C.byref(restart), C.byref(t), r, C.byref(ang)) == 0
        assert restart.value == b"c" and sfs.value == 2 and ang.value == 2 and t.value == pytest.approx(0.4)
        lists.append(l)
        k.restrt, k.scatt_framestart, k.time_now_start, k.restart_list = b"c", sfs.value, t.value, C.pointer(l)
        k.framestart, k.frm2 = fs.value, frm2.value
        k.seeds_drawn = 0
    # the continued run draws new seeds (as the reference's restarted generator does), so compare clocks and invariants, not photons
    cont = run(c, 2, 0, [str(tmp_path / "c0") + "/", str(tmp_path / "c1") + "/"])
    for r in range(2):
        assert len(cont[r]) == len(whole[r]) and c[r].time_now == pytest.approx(0.8) and a[r].time_now == pytest.approx(0.8)
        assert np.array_equal(cont[r]["weight"], whole[r]["weight"])
        assert (cont[r]["num_scatt"] >= 0).all() and c[r].frame_scatt_cnt_total > 0
        assert np.allclose(np.sqrt(cont[r]["p1"] ** 2 + cont[r]["p2"] ** 2 + cont[r]["p3"] ** 2), cont[r]["p0"], rtol=1e-12)
    libc = C.CDLL(None)
    for l in lists:
        libc.free(C.c_void_p(l.photons))


@pytest.mark.parametrize("late", [0, 1], ids=["all-join-at-0", "two-join-at-1"])
@pytest.mark.parametrize("sync_output", [0, 1], ids=["writer-thread", "sync-output"])
def test_two_hydro_frames_per_launch_write_the_files_of_one_frame_per_launch(hip, tmp_path, sync_output, late):
    """mcrat_host_pool_config.stage_ctx: frame F + 1 staged on a second context for what the photons can reach from frame F, every list through both
    frames in ONE launch (a list that is through F goes on in F + 1 while others are still in F: mcrat.c:566-934, the ranks' own frame loops), each
    frame's files written from the lists as that frame left them.  Four ranks of two injection radii that all join at frame 0 and run to frame 5:
    launches (0,1) (2,3) (4,5); with two of them joining at frame 1: (0) alone -- a rank joins at 1 --, then (1,2) (3,4) and (5).  Against the same run
    one frame per launch: every mc_proc dataset and every checkpoint record equal (but for
    nearest_block_index, which numbers the cells of whatever slab was read -- here a wider one), the same scattering counts in the logs."""
    from mcrat_amd.host import binding as B
    host, h5 = B.host(), B.host_h5()
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    jet = hip.Engine.outflow(3, lumi=2e53, theta_j=0.1)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    R, LASTF = 4, 5

    def run(tag, staged):
        ranks = (B.HostRank * R)()
        dirs, logs = [], []
        for r, k in enumerate(ranks):
            k.myid, k.angle_id, k.angle_procs = r, r, R
            k.theta_jmin_thread, k.theta_jmax_thread, k.ph_weight_suggest = 0.02 * (r % 2), 0.02 * (r % 2) + 0.05, 1e50
            k.inj_radius = 1e12 if r < 2 else 1.2e12
            k.framestart = k.frm2 = late * (r % 2)
            k.rng_seed, k.rng_stream = 4242, r
            d = str(tmp_path / ("%s_%d" % (tag, r))) + "/"
            os.makedirs(d)
            k.mc_dir = d.encode()
            k.fPtr = libc.fopen((d + "mc_output_%d.log" % r).encode(), b"a")
            dirs.append(d)
            logs.append(k.fPtr)
        pool = hip.Engine(synth.TWO, synth.SPHERICAL, 1)
        stage = hip.Engine(synth.TWO, synth.SPHERICAL, 1) if staged else None
        reads = []

        def reader(user, ctx, frame, slab):
            s = _slab_dict(slab.contents)
            on_stage = staged and ctx == stage.ctx.value
            reads.append((frame, s["ph_inj_switch"], bool(on_stage), s["min_r"], s["max_r"]))
            (stage if on_stage else pool).ingest(raw, s, jet)
            return 0
        cfg = B.PoolConfig()
        cfg.fps, cfg.last_frm = FPS, LASTF
        for k, v in DOM.items():
            getattr(cfg, k)[0], getattr(cfg, k)[1] = v
        cfg.spect, cfg.min_photons, cfg.max_photons = b"b", 300, 900
        cfg.get_hydro = B.GET_HYDRO(reader)
        cfg.write_checkpoints = 1
        if h5 is not None:
            cfg.print_photons = C.cast(h5.mcrat_host_print_photon_arrays, C.c_void_p).value
        cfg.comv_switch, cfg.stokes_switch, cfg.save_type = 1, 1, 1
        cfg.mode, cfg.sync_output, cfg.output_threads = hip.MODE_EXACT, sync_output, 2
        if staged:
            cfg.stage_ctx = stage.ctx.value
        assert host.mcrat_host_run_ranks(pool.ctx, ranks, R, C.byref(cfg)) == 0
        for f in logs:
            libc.fclose(f)
        out = dict(dirs=dirs, reads=reads, launches=cfg.launches, two=cfg.two_frame_launches, scatt=[k.frame_scatt_cnt_total for k in ranks],
                   t=[k.time_now for k in ranks], n=[k.num_photons for k in ranks])
        pool.close()
        if stage is not None:
            stage.close()
        return out
    one = run("one", False)
    two = run("two", True)
    assert one["launches"] == 6 and one["two"] == 0
    staged_reads = [(f, st) for f, inj, st, lo, hi in two["reads"] if not inj]
    if late:
        assert two["launches"] == 4 and two["two"] == 2
        assert staged_reads == [(0, False), (1, False), (2, True), (3, False), (4, True), (5, False)]
    else:
        assert two["launches"] == 3 and two["two"] == 3
        assert staged_reads == [(0, False), (1, True), (2, False), (3, True), (4, False), (5, True)]
    reach = 2.99792458e10 / FPS
    plain = [x for x in two["reads"] if not x[1]]
    for a_, b_ in zip(plain, plain[1:]):
        if b_[2]:                                                                   # a staged read: the pool's slab of the frame before, widened
            assert b_[3] == pytest.approx(max(a_[3] - reach, 0)) and b_[4] == pytest.approx(a_[4] + reach) and b_[0] == a_[0] + 1 and not a_[2]
    assert one["scatt"] == two["scatt"] and sum(one["scatt"]) > 200 and one["t"] == two["t"] and one["n"] == two["n"]
    head = 4 + 1 + 4 + 4
    for r in range(R):
        a, b = one["dirs"][r], two["dirs"][r]
        for name in ("mc_chkpt_%d.dat_old" % r,):
            x, y = open(a + name, "rb").read(), open(b + name, "rb").read()
            assert x[:head] == y[:head] and len(x) == len(y)
            ra, rb = np.frombuffer(x[head:], dtype=hip.PHOTON_DTYPE), np.frombuffer(y[head:], dtype=hip.PHOTON_DTYPE)
            for col in ra.dtype.names:
                if col != "nearest_block_index":
                    assert np.array_equal(ra[col], rb[col], equal_nan=ra[col].dtype.kind == "f"), (r, col)
        la, lb = open(a + "mc_output_%d.log" % r).read(), open(b + "mc_output_%d.log" % r).read()
        pick = lambda t: [l for l in t.splitlines() if "scatterings in this frame" in l or "Working on" in l or "propagating" in l]
        assert pick(la) == pick(lb) and len(pick(la)) == 3 * (LASTF + 1 - late * (r % 2)) + 1     # (+ the injection frame's own "Working on Frame")
        if h5 is not None:
            for F in range(late * (r % 2), LASTF + 1):
                for name, is_char in (("P0", 0), ("P3", 0), ("R0", 0), ("R2", 0), ("COMV_P0", 0), ("S1", 0), ("S3", 0), ("NS", 0), ("PW", 0), ("PT", 1)):
                    got = []
                    for path in (a, b):
                        n = C.c_int()
                        buf = np.empty(one["n"][r], dtype="S1" if is_char else np.float64)
                        assert h5.mcrat_host_h5_read((path + "mc_proc_%d.h5" % r).encode(), str(F).encode(), name.encode(), is_char, buf.ctypes.data,
                                                     one["n"][r], C.byref(n)) == 0
                        got.append(buf[:n.value].copy())
                    bad = np.nonzero(got[0] != got[1])[0] if len(got[0]) == len(got[1]) else None
                    assert len(got[0]) == len(got[1]) > 0 and np.array_equal(got[0], got[1]), (r, F, name, len(got[0]), len(got[1]), one["n"][r],
                                                                                                 None if bad is None else (len(bad), bad[:5], bad[-5:]))
