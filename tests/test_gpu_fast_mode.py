"""MCRAT_HIP_MODE_FAST (mcrat_hip_propagate_frame_mode): every photon through the frame on
its own clock with per-photon keyed random numbers.  It is statistically, not sequence-, equivalent to the event-driven loop of
mcrat.c:761-851, so it is held against the EXACT mode (itself parity-tested against the oracle) through the distribution gates
BASELINE.md names -- scatterings per photon, the energy spectrum, the Stokes parameters Q and U -- within Monte-Carlo error, and through
what must hold exactly: the clock, conservation, null slots, determinism."""
import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _run(hip, frame, ph, cfg, mode, seed, windows=0, **kw):
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], **kw)
    e.set_hydro(frame)
    e.set_photons(ph)
    rem = 1.0 / frame["fps"]
    if mode == "exact":
        tn, st = e.propagate_frame(3.0, rem, seed)
    else:
        tn, st = e.propagate_frame_fast(3.0, rem, seed, windows)
    out = e.get_photons()
    e.close()
    return tn, st, out


def _moments(ph0, o):
    n = len(o["p0"])
    ns = o["num_scatt"] - ph0["num_scatt"]
    loge = np.log(o["p0"])
    return dict(ns=(ns.mean(), ns.std() / np.sqrt(n)), loge=(loge.mean(), loge.std() / np.sqrt(n)),
                q=(o["s1"].mean(), o["s1"].std() / np.sqrt(n)), u=(o["s2"].mean(), o["s2"].std() / np.sqrt(n)))


def _spectrum_chi2(ph0, a, b, bins=24):
    """chi^2 per bin between the energy spectra of the photons that scattered"""
    la, lb = np.log(a["p0"][a["num_scatt"] > ph0["num_scatt"]]), np.log(b["p0"][b["num_scatt"] > ph0["num_scatt"]])
    lo, hi = min(la.min(), lb.min()), max(la.max(), lb.max())
    ha, _ = np.histogram(la, bins=bins, range=(lo, hi))
    hb, _ = np.histogram(lb, bins=bins, range=(lo, hi))
    ok = (ha + hb) >= 20
    assert ok.sum() >= 6
    return (((ha - hb) ** 2) / np.maximum(1, ha + hb))[ok].sum() / ok.sum()


@pytest.mark.parametrize("case", ["cfg2-cylindrical-stokes", "cfg3-spherical-stokes", "cfg2-cylindrical-hot"])
def test_fast_mode_agrees_with_exact_mode_in_distribution(hip, case):
    if case == "cfg2-cylindrical-stokes":
        frame, ph, cfg = synth.config2(n_photons=200_000, nzc=32, stokes=1, lumi=3e52)
    elif case == "cfg3-spherical-stokes":
        frame, ph, cfg = synth.config3(n_photons=150_000, nr=1024, nth=256, stokes=1, lumi=3e52)
    else:                                                         # T' > 1e7 K: the Maxwell-Juettner sampler, Klein-Nishina rejections
        frame, ph, cfg = synth.config2(n_photons=100_000, nzc=32, stokes=0, lumi=1e54, r_inj=1e11)
    _, st_e, ex = _run(hip, frame, ph, cfg, "exact", 777, virtual_rank_photons=1000)
    _, st_f, fa = _run(hip, frame, ph, cfg, "fast", 778)
    n = len(ph["p0"])
    assert st_e.frame_scatt_cnt > 0.1 * n                          # a frame in which the gates mean something
    assert int((fa["num_scatt"] - ph["num_scatt"]).sum()) == st_f.frame_scatt_cnt
    me, mf = _moments(ph, ex), _moments(ph, fa)
    for k in ("ns", "loge") + (("q", "u") if cfg["stokes"] else ()):
        z = (me[k][0] - mf[k][0]) / np.hypot(me[k][1], mf[k][1])
        assert abs(z) < 4.0, (case, k, me[k], mf[k], z)            # within Monte-Carlo error
    assert _spectrum_chi2(ph, ex, fa) < 2.0
    # the rejection rate of the Klein-Nishina test is a property of the frame, not of the mode
    re, rf = st_e.kn_rejections / st_e.frame_scatt_cnt, st_f.kn_rejections / st_f.frame_scatt_cnt
    assert abs(re - rf) < 4 * np.sqrt((re + rf + 1e-9) / st_e.frame_scatt_cnt) + 1e-4
    # and an exact run with another seed is as far from the first as FAST is (the gates are not vacuous)
    _, _, ex2 = _run(hip, frame, ph, cfg, "exact", 999, virtual_rank_photons=1000)
    assert not np.array_equal(ex2["p0"], ex["p0"])
    assert _spectrum_chi2(ph, ex, ex2) < 2.0


def test_fast_mode_bookkeeping(hip):
    frame, ph, cfg = synth.config2(n_photons=60_000, nzc=16, stokes=1, lumi=1e52)
    n = len(ph["p0"])
    ph = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ph.items()}
    null = np.arange(0, n, 97)
    ph["weight"][null] = 0.0                                        # null slots (setNullPhoton, photons.c:210-250): no weight, no cell
    ph["type"][null] = b"N"
    ph["nearest_block_index"][null] = -1
    far = np.arange(5, n, 211)                                      # photons about to leave the domain
    ph["r2"][far] = 2.49999e13
    ph["p3"][far] = np.abs(ph["p0"][far]); ph["p1"][far] = 0; ph["p2"][far] = 0
    tn, st, out = _run(hip, frame, ph, cfg, "fast", 5, windows=4)
    rem = 1.0 / frame["fps"]
    assert tn == 3.0 + rem and st.remaining_time == 0.0 and st.time_now == tn
    assert np.array_equal(out["weight"], ph["weight"]) and np.array_equal(out["type"], ph["type"])
    for k in ("r0", "r1", "r2", "p0", "p1", "p2", "p3"):
        assert np.array_equal(out[k][null], ph[k][null]), k        # untouched
    assert (out["num_scatt"] >= ph["num_scatt"]).all()
    assert int((out["num_scatt"] - ph["num_scatt"]).sum()) == st.frame_scatt_cnt > 1000
    nrm = np.sqrt(out["p1"] ** 2 + out["p2"] ** 2 + out["p3"] ** 2)
    assert np.allclose(nrm, out["p0"], rtol=1e-12, atol=0)
    assert (out["s0"] == 1).all() and (out["s1"] ** 2 + out["s2"] ** 2 + out["s3"] ** 2 <= 1 + 1e-9).all()
    gone = np.setdiff1d(far, null)
    assert (out["nearest_block_index"][gone] == -1).all()           # mclib.c:592
    # a photon that never scattered flew straight for the whole frame
    same = (out["num_scatt"] == ph["num_scatt"]) & (ph["weight"] != 0)
    d = synth.C_LIGHT * rem
    for r, p in (("r0", "p1"), ("r1", "p2"), ("r2", "p3")):
        assert np.allclose(out[r][same], ph[r][same] + d * ph[p][same] / ph["p0"][same], rtol=1e-12, atol=1e-3)
    assert st.photon_steps >= int((ph["weight"] != 0).sum()) and st.iterations >= 4


def test_fast_mode_is_deterministic_and_ignores_the_list_structure(hip):
    frame, ph, cfg = synth.config3(n_photons=40_000, nr=512, nth=128, stokes=1, lumi=3e52)
    a = _run(hip, frame, ph, cfg, "fast", 11, windows=8)
    b = _run(hip, frame, ph, cfg, "fast", 11, windows=8)
    c = _run(hip, frame, ph, cfg, "fast", 11, windows=8, virtual_rank_photons=1000)      # the lists do not matter to it
    d = _run(hip, frame, ph, cfg, "fast", 12, windows=8)
    for k in a[2]:
        assert np.array_equal(a[2][k], b[2][k]) and np.array_equal(a[2][k], c[2][k]), k
    assert a[1].frame_scatt_cnt == b[1].frame_scatt_cnt == c[1].frame_scatt_cnt
    assert not np.array_equal(a[2]["p0"], d[2]["p0"])
    # more windows: more re-locations and draws, the same physics
    e = _run(hip, frame, ph, cfg, "fast", 11, windows=32)
    assert e[1].photon_steps > a[1].photon_steps
    me, mf = _moments(ph, a[2]), _moments(ph, e[2])
    assert abs(me["ns"][0] - mf["ns"][0]) < 4 * np.hypot(me["ns"][1], mf["ns"][1])


def test_mode_argument(hip):
    import ctypes as C
    frame, ph, cfg = synth.config1(n_photons=3000)
    rem = 1.0 / frame["fps"]
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    tn, st = C.c_double(0.0), hip.FrameStats()
    assert e.lib.mcrat_hip_propagate_frame_mode(e.ctx, C.byref(tn), rem, 42, hip.MODE_EXACT, 0, C.byref(st)) == 0
    via_mode = e.get_photons()
    e.set_photons(ph)
    tn2, st2 = e.propagate_frame(0.0, rem, 42)
    plain = e.get_photons()
    assert tn.value == tn2 and st.frame_scatt_cnt == st2.frame_scatt_cnt
    for k in plain:
        assert np.array_equal(plain[k], via_mode[k]), k            # MODE_EXACT is mcrat_hip_propagate_frame
    assert e.lib.mcrat_hip_propagate_frame_mode(e.ctx, C.byref(tn), rem, 42, 7, 0, C.byref(st)) == -1      # MCRAT_HIP_EINVAL
    e.close()
    cs = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], cyclosynchrotron=1)
    cs.set_hydro(frame)
    cs.set_photons(ph)
    with pytest.raises(hip.McratHipError, match="cyclo-synchrotron"):
        cs.propagate_frame_fast(0.0, rem, 1)
    cs.close()


def test_fast_mode_on_a_rank_pool(hip):
    """the lists of a rank pool in one launch: every list's photons advance by the frame, closed windows stay empty"""
    frame, ph, cfg = synth.config2(n_photons=30_000, nzc=16, stokes=0, lumi=3e52)
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(4, 12_000)
    cuts = [(0, 9_000), (9_000, 20_500), (20_500, 30_000)]                  # list 3 stays empty
    views = [pool.pool_rank(r, 40 + r) for r in range(4)]
    for v, (lo, hi) in zip(views, cuts):
        v.set_photons({k: (a[lo:hi].copy() if isinstance(a, np.ndarray) else a) for k, a in ph.items()})
    rem = 1.0 / frame["fps"]
    tn, st = pool.propagate_frame_fast(1.0, rem, 21)
    assert tn == 1.0 + rem and st.frame_scatt_cnt > 3000
    total = 0
    for v, (lo, hi) in zip(views, cuts):
        out = v.get_photons()
        assert len(out["p0"]) == hi - lo
        assert np.array_equal(out["weight"], ph["weight"][lo:hi])
        ns = out["num_scatt"] - ph["num_scatt"][lo:hi]
        assert (ns >= 0).all()
        total += int(ns.sum())
        moved = np.sqrt((out["r0"] - ph["r0"][lo:hi]) ** 2 + (out["r1"] - ph["r1"][lo:hi]) ** 2 + (out["r2"] - ph["r2"][lo:hi]) ** 2)
        assert np.allclose(moved[ns == 0], synth.C_LIGHT * rem, rtol=1e-9)
    assert total == st.frame_scatt_cnt
    assert pool.pool_summaries()[3].list_capacity == 0
    # a list of the pool is a context of its own: same stream, same seed -> the same photons bit for bit, whatever else shares the pool
    lo, hi = cuts[1]
    own = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=41)
    own.set_hydro(frame)
    own.set_photons({k: (a[lo:hi].copy() if isinstance(a, np.ndarray) else a) for k, a in ph.items()})
    own.propagate_frame_fast(1.0, rem, 21)
    a, b = own.get_photons(), views[1].get_photons()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    own.close()
    # per-list seeds, clocks and counters
    for v, (lo, hi) in zip(views, cuts):
        v.set_photons({k: (a[lo:hi].copy() if isinstance(a, np.ndarray) else a) for k, a in ph.items()})
    per = pool.pool_propagate_frames_fast([1, 0, 1, 0], [5, 6, 7, 8], [0.0, 0.0, 2.0, 0.0], [rem, rem, rem / 2, rem])
    assert per[0].time_now == rem and per[2].time_now == 2.0 + rem / 2 and per[0].frame_scatt_cnt > per[2].frame_scatt_cnt > 500
    untouched = views[1].get_photons()
    assert np.array_equal(untouched["r0"], ph["r0"][cuts[1][0]:cuts[1][1]]) and np.array_equal(untouched["num_scatt"], ph["num_scatt"][cuts[1][0]:cuts[1][1]])
    assert int((views[2].get_photons()["num_scatt"] - ph["num_scatt"][cuts[2][0]:cuts[2][1]]).sum()) == per[2].frame_scatt_cnt
    pool.close()


def _hot_table():
    """a smooth stand-in for thermal_hot_x_section.dat on the reference's grid (as tests/test_gpu_parity.py)"""
    i, j = np.meshgrid(np.arange(221), np.arange(81), indexing="ij")
    x = -12.0 + i * (18.0 / 220)
    y = -4.0 + j * (8.0 / 80)
    return -0.35 * np.log1p(np.exp(2.0 * (x + 0.5))) / np.log(10) - 0.02 * (y + 4.0) * (1 + 0.1 * np.tanh(x))


@pytest.mark.parametrize("case", ["table-cfg2", "2.5d-spherical", "3d-spherical", "3d-cartesian"])
def test_fast_mode_in_the_other_builds_of_the_kernels(hip, case):
    """the kernels exist per TAU_CALCULATION x DIMENSIONS (kernels*_d*.hip): FAST against EXACT in the TABLE build and in the 2.5-D and 3-D ones"""
    kw, tab = {}, None
    if case == "table-cfg2":
        frame, ph, cfg = synth.config2(n_photons=60_000, nzc=16, stokes=1, lumi=3e52)
        kw, tab = dict(tau_calculation=hip.TAU_TABLE), _hot_table()
    elif case == "2.5d-spherical":
        frame, ph, cfg = synth.config_25d(synth.SPHERICAL, n_photons=40_000, stokes=1, lumi=3e53)
    elif case == "3d-spherical":
        frame, ph, cfg = synth.config_3d(synth.SPHERICAL, n_photons=40_000)
    else:
        frame, ph, cfg = synth.config_3d_cartesian(n_photons=40_000)
    rem = 1.0 / frame["fps"]
    res = {}
    for mode in ("exact", "fast"):
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000 if mode == "exact" else 0, **kw)
        if tab is not None:
            e.set_hot_cross_section(tab)
        e.set_hydro(frame)
        e.set_photons(ph)
        if mode == "exact":
            _, st = e.propagate_frame(0.0, rem, 31)
        else:
            _, st = e.propagate_frame_fast(0.0, rem, 32, 16)
        res[mode] = (st, e.get_photons())
        e.close()
    st_e, ex = res["exact"]
    st_f, fa = res["fast"]
    assert st_e.frame_scatt_cnt > 2000, st_e.frame_scatt_cnt
    assert int((fa["num_scatt"] - ph["num_scatt"]).sum()) == st_f.frame_scatt_cnt
    me, mf = _moments(ph, ex), _moments(ph, fa)
    for k in ("ns", "loge", "q", "u"):
        z = (me[k][0] - mf[k][0]) / max(1e-300, np.hypot(me[k][1], mf[k][1]))
        assert abs(z) < 4.5, (case, k, me[k], mf[k], z)
    assert st_f.not_found == st_e.not_found == 0 or abs(st_f.not_found - st_e.not_found) <= 0.2 * max(st_e.not_found, 50)


def test_learnt_cadence_matches_the_exact_loop_on_a_dense_run_of_ten_million_events(hip):
    """fast_windows = 0: the context refreshes a photon's cell and optical depth as often per frame as a 1000-photon rank of the exact loop would have
    (the scatterings of the frame before per thousand photons, 8 ... 2048).  Three frames of the L = 1e54 jet, 1.9e7 scatterings: the mean number of
    scatterings per photon agrees with the exact loop's within 4 sigma of their Monte-Carlo errors -- and the gate is not vacuous: 8 fixed windows
    (round 2's default) miss it by more than twenty sigma."""
    n, frames = 1000000, 3
    frame, ph, cfg = synth.config2(n_photons=n, lumi=1e54)
    dt = 1.0 / frame["fps"]
    mean, err, events = {}, {}, {}
    for mode, windows in (("exact", None), ("auto", 0), ("fixed-8", 8)):
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000 if mode == "exact" else 0)
        e.set_hydro(frame)
        e.set_photons(ph)
        t, ev = 0.0, 0
        for f in range(frames):
            if mode == "exact":
                t, st = e.propagate_frame(t, (f + 1) * dt - t, 1000 + f)
            else:
                t, st = e.propagate_frame_fast(t, (f + 1) * dt - t, 1000 + f, windows)
            ev += st.frame_scatt_cnt
        ns = np.asarray(e.get_photons()["num_scatt"]) - np.asarray(ph["num_scatt"])
        e.close()
        mean[mode], err[mode], events[mode] = float(ns.mean()), float(ns.std() / np.sqrt(n)), ev
    assert events["exact"] > 1e7
    sigma = np.hypot(err["exact"], err["auto"])
    assert abs(mean["auto"] - mean["exact"]) < 4 * sigma, (mean, err)
    assert abs(mean["fixed-8"] - mean["exact"]) > 10 * sigma, (mean, err)


def test_the_learnt_cadence_belongs_to_the_list_not_to_the_pool(hip):
    """fast_windows <= 0 learns the refresh cadence from the frame before -- of THAT list: a dense list sharing a pool with a thin one must run its second
    frame exactly as it does alone in a context of its own (round 3 learnt one number from the summed scatterings of all adopted ranks: a rank's
    photons then depended on which other ranks the process had adopted), and a restarted run that hands the value back (mcrat_hip_fast_cadence)
    goes on as the uninterrupted one."""
    dense, ph_d, cfg = synth.config2(n_photons=6_000, nzc=16, stokes=0, lumi=1e53)
    rem = 1.0 / dense["fps"]
    # the photons as list 0 of a pool whose list 1 holds photons that hardly scatter on the same frame (far outside the jet's core)
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(dense)
    pool.pool_create(2, 6_144)
    v0, v1 = pool.pool_rank(0, 11), pool.pool_rank(1, 12)
    v0.set_photons(ph_d)
    quiet = {k: (a.copy() if isinstance(a, np.ndarray) else a) for k, a in ph_d.items()}
    quiet["r0"] = quiet["r0"] + 2e11                           # moved off the axis, beyond the jet's opening angle: far fewer scatterings per photon
    v1.set_photons(quiet)
    s1 = pool.pool_propagate_frames_fast([1, 1], [5, 6], [0.0, 0.0], [rem, rem])
    c0, c1 = v0.fast_cadence(), v1.fast_cadence()
    assert c0 == min(2048, max(8, int(1000.0 * s1[0].frame_scatt_cnt / 6000 + 0.5)))
    assert c1 == min(2048, max(8, int(1000.0 * s1[1].frame_scatt_cnt / 6000 + 0.5))) and c0 != c1
    pool.pool_propagate_frames_fast([1, 1], [7, 8], [rem, rem], [rem, rem])
    own = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=11)
    own.set_hydro(dense)
    own.set_photons(ph_d)
    own.propagate_frame_fast(0.0, rem, 5)
    assert own.fast_cadence() == c0
    mid = own.get_photons()
    own.propagate_frame_fast(rem, rem, 7)
    a, b = own.get_photons(), v0.get_photons()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # "restart": a fresh context, the interrupted run's photons and its cadence
    again = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=11)
    again.set_hydro(dense)
    again.set_photons(mid)
    assert again.fast_cadence() == 32 and again.fast_cadence(c0) == c0
    again.propagate_frame_fast(rem, rem, 7)
    c = again.get_photons()
    for k in a:
        assert np.array_equal(a[k], c[k]), k
    for e in (pool, own, again):
        e.close()
