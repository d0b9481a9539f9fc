"""Full-size checks (BASELINE.json configs[1] and [2] shapes: 10^6 photons on the 1 048 576-cell frames) through
size-independent properties -- the CPU oracle would need hours at these sizes (its cell search is the reference's
linear scan):

  * the clock: the loop ends exactly at the frame time, time_now advanced by the frame;
  * photons are conserved, stay null (|p| = p0), keep their weight and type; scattering counts only grow and add up
    to the frame's event counter;
  * every photon in the domain sits in the cell the device says it is in (closed-interval test on the host, on the
    position the cell was found for), and right after the forced pass its stored comoving 4-momentum is the Lorentz
    transform of its lab 4-momentum with that cell's velocity at the photon's azimuth;
  * Stokes vectors stay physical: I = 1, Q^2 + U^2 + V^2 <= 1;
  * the run is deterministic, and splitting it anywhere does not change a bit;
  * a single list and the same photons as 1000 independent lists agree on what does not depend on the clock:
    the forced pass locates every photon in the same cell.
"""
import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu

N = 1_000_000


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _invariants(frame, cfg, ph0, out, st, n_events, dt_last):
    n = len(ph0["p0"])
    assert len(out["p0"]) == n
    assert np.array_equal(out["weight"], ph0["weight"]) and np.array_equal(out["type"], ph0["type"])
    for k in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "comv_p0"):
        assert np.isfinite(out[k]).all(), k
    nrm = np.sqrt(out["p1"] ** 2 + out["p2"] ** 2 + out["p3"] ** 2)
    assert np.allclose(nrm, out["p0"], rtol=1e-12, atol=0)                      # zeroNorm, mclib.c:409
    assert (out["num_scatt"] >= ph0["num_scatt"]).all()
    assert int((out["num_scatt"] - ph0["num_scatt"]).sum()) == n_events
    # where the device says the photons are: nearest_block_index is the cell of the last findContainingHydroCell, i.e.
    # of the position BEFORE the last advance (mcrat.c:768 then :781/:841), for photons that have not scattered since
    idx = out["nearest_block_index"]
    inside = idx >= 0
    assert inside.mean() > 0.99
    ok = inside & (out["recalc_properties"] == 0)
    back = synth.C_LIGHT * dt_last / out["p0"]
    x, y, z = out["r0"] - out["p1"] * back, out["r1"] - out["p2"] * back, out["r2"] - out["p3"] * back
    if cfg["geometry"] == synth.SPHERICAL:
        a0 = np.sqrt(x ** 2 + y ** 2 + z ** 2)
        a1 = np.arccos(z / a0)
    else:
        a0, a1 = np.hypot(x, y), z
    c = idx[ok]
    assert (2 * np.abs(a0[ok] - frame["r0"][c]) - frame["r0_size"][c] <= 1e-9 * frame["r0_size"][c]).all()
    assert (2 * np.abs(a1[ok] - frame["r1"][c]) - frame["r1_size"][c] <= 1e-9 * frame["r1_size"][c]).all()
    # Stokes
    if cfg["stokes"]:
        assert (out["s0"] == 1).all()
        pol = out["s1"] ** 2 + out["s2"] ** 2 + out["s3"] ** 2
        assert (pol <= 1 + 1e-9).all() and pol.max() > 1e-6
    else:
        assert (out["s0"] == ph0["s0"]).all() and (out["s1"] == 0).all()


def _comoving_after_forced_pass(frame, cfg, out, dt_last):
    """right after the forced pass (mcrat.c:756: every photon located anew) the stored comoving 4-momentum is the
    Lorentz transform of the lab 4-momentum with the cell's velocity at the photon's azimuth (mclib.c:549-558)"""
    idx = out["nearest_block_index"]
    ok = (idx >= 0) & (out["recalc_properties"] == 0)
    assert ok.mean() > 0.99
    back = synth.C_LIGHT * dt_last / out["p0"]
    x, y = out["r0"] - out["p1"] * back, out["r1"] - out["p2"] * back
    sel = np.nonzero(ok)[0][:: max(1, int(ok.sum()) // 50000)]
    beta = synth.hydro_vector_to_cartesian(frame, idx[sel], np.arctan2(y[sel], x[sel]))
    p4 = np.stack([out["p0"][sel], out["p1"][sel], out["p2"][sel], out["p3"][sel]], axis=-1)
    comv = synth.lorentz_boost(beta, p4)
    for k, name in enumerate(("comv_p0", "comv_p1", "comv_p2", "comv_p3")):
        assert np.allclose(comv[:, k], out[name][sel], rtol=0, atol=2e-8 * np.abs(out["comv_p0"][sel]).max()) and \
               (np.abs(comv[:, k] - out[name][sel]) <= 2e-8 * np.abs(out["comv_p0"][sel])).all(), name


@pytest.mark.parametrize("which", ["cfg2-flash-cylindrical", "cfg3-pluto-spherical-stokes"])
def test_whole_frame_as_virtual_ranks_full_size(hip, which):
    if which.startswith("cfg2"):
        frame, ph, cfg = synth.config2(n_photons=N)
    else:
        frame, ph, cfg = synth.config3(n_photons=N)
    assert frame["num_elements"] == 1048576
    rem = 1.0 / frame["fps"]
    runs = []
    for split in (0, 3):
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.begin_frame(20261003, 7.0, rem)
        if split:
            e.run(1)                                           # the forced pass of every list
            dt1 = np.repeat([e.rank_stats(r).last_time_step for r in range(e.num_virtual_ranks())], 1000)[:N]
            _comoving_after_forced_pass(frame, cfg, e.get_photons(), dt1)
            e.run(split - 1)
        st = e.run(0)
        dt = np.repeat([e.rank_stats(r).last_time_step for r in range(e.num_virtual_ranks())], 1000)[:N]
        runs.append((st, e.get_photons(), dt))
        e.close()
    st, out, dt_last = runs[0]
    assert st.remaining_time == 0.0 and st.time_now == pytest.approx(7.0 + rem, rel=1e-14)
    assert st.frame_scatt_cnt > 1000 and st.not_found == 0
    _invariants(frame, cfg, ph, out, st, st.frame_scatt_cnt, dt_last)
    st2, out2, _ = runs[1]
    assert (st2.iterations, st2.frame_scatt_cnt, st2.num_photons_find_new_element) == (st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element)
    for k in out:
        assert np.array_equal(np.asarray(out[k]), np.asarray(out2[k]), equal_nan=(np.asarray(out[k]).dtype.kind == "f")), k


def test_one_list_full_size(hip):
    frame, ph, cfg = synth.config2(n_photons=N)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=250, use_graph=True)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(99, 0.0, 1.0 / frame["fps"])
    first = e.run(1)                                           # the forced pass
    after1 = e.get_photons()
    loc = after1["nearest_block_index"].copy()
    _comoving_after_forced_pass(frame, cfg, after1, np.full(N, first.last_time_step))
    st = e.run(1499)
    out = e.get_photons()
    assert st.iterations == 1500 and st.frame_scatt_cnt >= 1490
    assert st.time_now + st.remaining_time == pytest.approx(1.0 / frame["fps"], rel=1e-12)
    _invariants(frame, cfg, ph, out, st, st.frame_scatt_cnt, np.full(N, st.last_time_step))
    # graph replay and eager launches are the same run
    e2 = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=500, use_graph=False)
    e2.set_hydro(frame)
    e2.set_photons(ph)
    e2.begin_frame(99, 0.0, 1.0 / frame["fps"])
    st2 = e2.run(1500)
    out2 = e2.get_photons()
    assert st2.frame_scatt_cnt == st.frame_scatt_cnt and st2.time_now == st.time_now
    for k in out:
        assert np.array_equal(np.asarray(out[k]), np.asarray(out2[k]), equal_nan=(np.asarray(out[k]).dtype.kind == "f")), k
    # the same photons as 1000 independent lists: the forced pass finds the same cells
    r = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000)
    r.set_hydro(frame)
    r.set_photons(ph)
    r.begin_frame(99, 0.0, 1.0 / frame["fps"])
    r.run(1)
    assert np.array_equal(r.get_photons()["nearest_block_index"], loc)
    assert first.iterations == 1


def test_cfg3_at_its_full_ten_million_photons_as_a_rank_pool(hip):
    """BASELINE.json configs[2] at full size: 10^7 photons on the 1 048 576-cell spherical frame, Stokes on, as the rank pool bench.py --config cfg3
    runs (10 246 adopted ranks of ~976 photons, each with its own stream and seed): the invariants of this file on every list's photons, the
    per-list statistics adding up, and FAST mode on the same pool within Monte-Carlo error of the exact frame"""
    n = 10_000_000
    frame, ph, cfg = synth.config3(n_photons=n)
    assert frame["num_elements"] == 1048576 and len(ph["p0"]) == n
    rem = 1.0 / frame["fps"]
    per = 976
    n_lists = int(round(n / per))
    rng = np.random.default_rng(3)
    lens = np.full(n_lists, per) + rng.integers(-40, 41, n_lists)
    lens[-1] += n - int(lens.sum())
    offs = np.concatenate([[0], np.cumsum(lens)])
    assert offs[-1] == n and lens.min() > 0
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(n_lists, int(lens.max()))
    views = []
    for r in range(n_lists):
        v = pool.pool_rank(r, 1000 + r)
        v.set_photons({k: (a[offs[r]:offs[r + 1]] if isinstance(a, np.ndarray) else a) for k, a in ph.items()})
        views.append(v)
    pool.snapshot_photons()
    pool.begin_frame(20261004, 2.0, rem)
    st = pool.run(0)
    assert st.remaining_time == 0.0 and st.not_found == 0 and st.frame_scatt_cnt > 50_000
    summ = pool.pool_summaries()
    assert sum(s.list_capacity for s in summ) == n
    # the pool's photons, list by list, in the order they went in
    cols = {k: [] for k in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3", "num_scatt", "weight", "type", "comv_p0")}
    for v in views[:: max(1, n_lists // 400)]:                        # every 25th list in full (the copies are what takes time here)
        o = v.get_photons()
        for k in cols:
            cols[k].append(o[k])
    sel = np.concatenate([np.arange(offs[r], offs[r + 1]) for r in range(0, n_lists, max(1, n_lists // 400))])
    got = {k: np.concatenate(v) for k, v in cols.items()}
    assert np.array_equal(got["weight"], ph["weight"][sel]) and np.array_equal(got["type"], ph["type"][sel])
    for k in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "comv_p0"):
        assert np.isfinite(got[k]).all(), k
    nrm = np.sqrt(got["p1"] ** 2 + got["p2"] ** 2 + got["p3"] ** 2)
    assert np.allclose(nrm, got["p0"], rtol=1e-12, atol=0)
    assert (got["num_scatt"] >= ph["num_scatt"][sel]).all()
    assert (got["s0"] == 1).all() and (got["s1"] ** 2 + got["s2"] ** 2 + got["s3"] ** 2 <= 1 + 1e-9).all()
    per_list = np.array([pool.rank_stats(r).frame_scatt_cnt for r in range(0, n_lists, max(1, n_lists // 400))])
    ns_exact = got["num_scatt"] - ph["num_scatt"][sel]
    assert int(ns_exact.sum()) == int(per_list.sum())
    # deterministic: the same frame again from the snapshot, bit for bit
    pool.restore_photons()
    pool.begin_frame(20261004, 2.0, rem)
    st2 = pool.run(0)
    assert (st2.iterations, st2.frame_scatt_cnt, st2.num_photons_find_new_element) == (st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element)
    again = views[0].get_photons()
    first = {k: got[k][: lens[0]] for k in ("p0", "r0", "s1")}
    for k in first:
        assert np.array_equal(again[k], first[k]), k
    # FAST mode on the same 10^7 photons: the same number of scatterings within Monte-Carlo error
    pool.restore_photons()
    _, stf = pool.propagate_frame_fast(2.0, rem, 77)
    assert abs(stf.frame_scatt_cnt - st.frame_scatt_cnt) < 6 * np.sqrt(st.frame_scatt_cnt) + 1e-3 * st.frame_scatt_cnt
    pool.close()


def test_cfg4_share_of_one_gpu_as_a_rank_pool_and_through_the_frame_queue(hip):
    """BASELINE.json configs[3] (10^8 photons on 8 GPUs, the cfg2 frame replicated) is 1.25e7 photons per GPU: ~12 800 adopted ranks on the 1 048 576-cell
    cylindrical frame.  One GPU's share at full size: the invariants of this file on a sample of lists, the per-list counters adding up, and the
    frame queue (two frames of every list in ONE launch, mcrat_hip_pool_run_frames) bit-identical to one launch per frame."""
    n = 12_500_000
    frame, ph, cfg = synth.config2(n_photons=n)
    assert frame["num_elements"] == 1048576 and len(ph["p0"]) == n
    rem = 1.0 / frame["fps"]
    per = 976
    n_lists = int(round(n / per))
    rng = np.random.default_rng(4)
    lens = np.full(n_lists, per) + rng.integers(-40, 41, n_lists)
    lens[-1] += n - int(lens.sum())
    offs = np.concatenate([[0], np.cumsum(lens)])
    assert offs[-1] == n and lens.min() > 0 and n_lists > 12_000
    recs = synth.photons_to_aos(ph, hip.PHOTON_DTYPE)

    def make():
        pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
        pool.set_hydro(frame)
        pool.pool_create(n_lists, int(lens.max()))
        for r in range(n_lists):
            pool.pool_rank(r, 5000 + r)
        pool.pool_set_photons(list(range(n_lists)), [recs[offs[r]:offs[r + 1]] for r in range(n_lists)])
        return pool
    pool = make()
    seeds = np.array([[77 + 13 * r for r in range(n_lists)], [900_001 + 7 * r for r in range(n_lists)]], dtype=np.uint64)
    # frame by frame, every list its own seed and the clock carried on
    totals = []
    t_now = np.zeros(n_lists)
    for f in range(2):
        t_rem = (f + 1) / frame["fps"] - t_now
        o, sd, t, rm = (np.ones(n_lists, dtype=np.int32), seeds[f], t_now.copy(), t_rem.copy())
        pool._check(pool.lib.mcrat_hip_pool_begin_frames(pool.ctx, o.ctypes.data_as(hip.C.POINTER(hip.C.c_int)), sd.ctypes.data_as(hip.C.POINTER(hip.C.c_uint64)),
                                                        t.ctypes.data_as(hip.C.POINTER(hip.C.c_double)), rm.ctypes.data_as(hip.C.POINTER(hip.C.c_double))), "pool_begin_frames")
        st = pool.run(0)
        assert st.remaining_time == 0.0 and st.not_found == 0
        totals.append((st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element))
        per_list = (hip.FrameStats * n_lists)()
        pool._check(pool.lib.mcrat_hip_pool_frame_stats(pool.ctx, per_list), "pool_frame_stats")
        t_now = np.array([s.time_now for s in per_list])
        assert sum(s.frame_scatt_cnt for s in per_list) == st.frame_scatt_cnt
    assert totals[0][1] > 60_000
    sample = list(range(0, n_lists, max(1, n_lists // 300)))
    got = {r: pool.views[r].get_photons() for r in sample}
    for r, o in got.items():
        lo, hi = offs[r], offs[r + 1]
        assert np.array_equal(o["weight"], ph["weight"][lo:hi]) and np.array_equal(o["type"], ph["type"][lo:hi])
        nrm = np.sqrt(o["p1"] ** 2 + o["p2"] ** 2 + o["p3"] ** 2)
        assert np.isfinite(o["p0"]).all() and np.allclose(nrm, o["p0"], rtol=1e-12, atol=0)
        assert (o["num_scatt"] >= ph["num_scatt"][lo:hi]).all()
    pool.close()
    # the same two frames of every list in ONE launch
    q = make()
    frame_end = np.array([[(f + 1) / frame["fps"]] * n_lists for f in range(2)])
    stq = q.pool_run_frames(np.ones((2, n_lists), dtype=np.int32), seeds, np.zeros((2, n_lists)), frame_end.copy(), frame_end=frame_end, chain_clock=True)
    for f in range(2):
        assert (sum(s.iterations for s in stq[f]), sum(s.frame_scatt_cnt for s in stq[f]), sum(s.num_photons_find_new_element for s in stq[f])) == totals[f]
    for r in sample:
        o = q.views[r].get_photons()
        for k in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "comv_p0", "num_scatt", "nearest_block_index"):
            assert np.array_equal(o[k], got[r][k], equal_nan=True), (r, k)
    q.close()
