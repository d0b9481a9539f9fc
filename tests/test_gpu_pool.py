"""Rank pool (mcrat_hip_pool_*): R independent photon lists of different lengths -- the reference's MPI ranks, each with its
own Poisson-sized list, generator and clock (mcrat.c:99-103,139-164,457-479,701; mclib.c:87-136) -- in ONE context, all lists
propagated by one launch.  Every list must equal the oracle run on that list alone (its own seed, stream, clock), and the
view of a list must behave like a context of its own for the per-list entry points.  Through the C ABI."""
import numpy as np
import pytest

from mcrat_amd import synth
from tests.test_gpu_parity import FLOAT_FIELDS, INT_FIELDS, _compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _cut(ph, lo, hi, n):
    return {k: (v[lo:hi].copy() if isinstance(v, np.ndarray) and len(v) == n else v) for k, v in ph.items()}


def _lists(ph, lens):
    n, out, lo = len(ph["p0"]), [], 0
    for m in lens:
        out.append(_cut(ph, lo, lo + m, n))
        lo += m
    assert lo <= n
    return out


def _hot_table():
    """a smooth stand-in for thermal_hot_x_section.dat on the reference's grid (as tests/test_gpu_parity.py)"""
    i, j = np.meshgrid(np.arange(221), np.arange(81), indexing="ij")
    x = -12.0 + i * (18.0 / 220)
    y = -4.0 + j * (8.0 / 80)
    return -0.35 * np.log1p(np.exp(2.0 * (x + 0.5))) / np.log(10) - 0.02 * (y + 4.0) * (1 + 0.1 * np.tanh(x))


@pytest.mark.parametrize("case", ["cfg2-stokes", "cfg3-stokes-long", "cfg1-global", "cfg2-table", "2.5d-cylindrical", "3d-spherical", "3d-cartesian"])
def test_unequal_lists_equal_independent_oracle_runs(hip, oracle, case, monkeypatch):
    lens = [137, 1000, 512, 999, 3, 64, 700, 1024]
    kw, okw = {}, {}
    if case == "cfg2-table":                          # the TAU_CALCULATION == TABLE build of the kernels (kernels_table_d0.hip)
        frame, ph, cfg = synth.config2(n_photons=sum(lens), nzc=8, stokes=1, lumi=1e54)
        kw, okw = dict(tau_calculation=hip.TAU_TABLE), dict(hot_table=_hot_table())
    elif case == "2.5d-cylindrical":                  # kernels_d1.hip
        frame, ph, cfg = synth.config_25d(synth.CYLINDRICAL, n_photons=sum(lens))
    elif case == "3d-spherical":                      # kernels_d2.hip
        frame, ph, cfg = synth.config_3d(synth.SPHERICAL, n_photons=sum(lens))
    elif case == "3d-cartesian":
        frame, ph, cfg = synth.config_3d_cartesian(n_photons=sum(lens))
    elif case == "cfg2-stokes":
        frame, ph, cfg = synth.config2(n_photons=sum(lens), nzc=8, stokes=1, lumi=1e54)
    elif case == "cfg3-stokes-long":
        lens[5] = 1500                                 # one list beyond what is kept in LDS: every list takes the HBM/L2 path
        frame, ph, cfg = synth.config3(n_photons=sum(lens), nr=256, nth=128, lumi=1e54)
    else:
        frame, ph, cfg = synth.config1(n_photons=sum(lens), n0=32, n1=32)
        monkeypatch.setenv("MCRAT_HIP_NO_LDS_LISTS", "1")
    subs = _lists(ph, lens)
    R = len(lens)
    seeds = [1000 + 17 * r for r in range(R)]
    streams = [3, 99, 4, 4, 7, 0, 12345, 8][:R]        # two ranks may even share a stream: their seeds differ
    t0 = [2.0 + 0.01 * r for r in range(R)]            # every rank its own clock
    rem = [0.2 - 0.01 * r for r in range(R)]
    passes = 250

    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], **kw)
    if "hot_table" in okw:
        pool.set_hot_cross_section(okw["hot_table"])
    pool.set_hydro(frame)
    pool.pool_create(R, 1600)
    views = []
    for r in range(R):
        if r == 4:
            continue                                   # rank 4 joins late (below)
        v = pool.pool_rank(r, streams[r])
        v.set_photons(subs[r])
        v.begin_frame(seeds[r], t0[r], rem[r])
        views.append(v)
    tot = pool.run(passes)

    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], **okw)
    sums = dict(it=0, sc=0)
    for r in range(R):
        if r == 4:
            continue
        P = oracle.OraclePhotons(synth.photons_to_aos(subs[r], oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seeds[r], time_now=t0[r], remaining_time=rem[r], max_iterations=passes,
                                               stream=streams[r])
        v = pool.pool_rank(r, streams[r])
        st = v.frame_statistics()
        assert st.iterations == rst.iterations
        assert st.frame_scatt_cnt == rst.frame_scatt_cnt
        assert st.kn_rejections == rst.kn_rejections
        assert st.num_photons_find_new_element == rst.num_photons_find_new_element
        assert st.last_scattered_index == rst.last_scattered_index          # list-local, as in a context of its own
        assert st.time_now == pytest.approx(rtn, rel=1e-12) and st.remaining_time == pytest.approx(rrem, rel=1e-9, abs=1e-300)
        _compare(v.get_photons(), P.aos)
        sums["it"] += rst.iterations
        sums["sc"] += rst.frame_scatt_cnt
    assert (tot.iterations, tot.frame_scatt_cnt) == (sums["it"], sums["sc"])
    assert sums["sc"] > 0

    # a rank that joins later (its own injection frame): only its list runs now, the others are untouched
    before = {r: pool.pool_rank(r, streams[r]).get_photons() for r in (0, 7)}
    v4 = pool.pool_rank(4, streams[4])
    v4.set_photons(subs[4])
    v4.begin_frame(seeds[4], t0[4], rem[4])
    pool.run(passes)
    P = oracle.OraclePhotons(synth.photons_to_aos(subs[4], oracle.PHOTON_DTYPE))
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=seeds[4], time_now=t0[4], remaining_time=rem[4], max_iterations=passes, stream=streams[4])
    assert v4.frame_statistics().iterations == rst.iterations
    _compare(v4.get_photons(), P.aos)
    for r in (0, 7):
        st = pool.pool_rank(r, streams[r]).frame_statistics()
        if st.remaining_time > 0:
            continue                                   # unfinished lists went on: not comparable
        after = pool.pool_rank(r, streams[r]).get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(before[r][k], after[k], equal_nan=True), (r, k)
    pool.close()


def test_pool_lists_are_bitwise_single_list_contexts_in_rank_mode(hip):
    """whole frames: list r of a pool == list r of a fixed-size virtual-rank context == bitwise, when lengths, seed and streams agree"""
    n, per, R = 6000, 1000, 6
    frame, ph, cfg = synth.config2(n_photons=n, nzc=8, lumi=1e53)
    rem = 1.0 / frame["fps"]
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=10, virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(77, 0.0, rem)
    st = e.run(0)
    ref = e.get_photons()
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(R, per)
    for r in range(R):
        v = pool.pool_rank(r, 10 + r)
        v.set_photons(_cut(ph, r * per, (r + 1) * per, n))
    pool.begin_frame(77, 0.0, rem)                     # the pool's own begin_frame: every list, one seed and clock
    tot = pool.run(0)
    assert (tot.iterations, tot.frame_scatt_cnt, tot.num_photons_find_new_element) == (st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element)
    assert tot.frame_scatt_cnt > 0 and tot.remaining_time == 0.0
    for r in range(R):
        out = pool.pool_rank(r, 10 + r).get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(out[k], np.asarray(ref[k])[r * per:(r + 1) * per], equal_nan=True), (r, k)
    e.close()
    pool.close()


def test_view_entry_points_and_summaries(hip, oracle):
    """a view is a context of its own for the per-list entry points: reductions, printPhotons' columns, checkpoint records;
    mcrat_hip_pool_summaries gives the same numbers for every list in one launch"""
    lens = [300, 1100, 17]
    frame, ph, cfg = synth.config2(n_photons=sum(lens), nzc=8, stokes=1, lumi=1e54)
    ph["weight"][5] = 0.0                              # a photon printPhotons skips
    subs = _lists(ph, lens)
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(4, 1200)                          # rank 3 never gets a list
    for r in range(3):
        v = pool.pool_rank(r, r)
        v.set_photons(subs[r])
        v.begin_frame(5 + r, 0.0, 0.05)
    pool.run(0)
    summ = pool.pool_summaries()
    assert summ[3].list_capacity == 0
    for r in range(3):
        v = pool.pool_rank(r, r)
        single = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=r, virtual_rank_photons=lens[r])
        single.set_hydro(frame)
        single.set_photons(subs[r])
        single.begin_frame(5 + r, 0.0, 0.05)
        single.run(0)
        a, b = v.get_photons(), single.get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(a[k], b[k], equal_nan=True), (r, k)
        assert v.ph_minmax() == single.ph_minmax()
        assert v.scatt_stats() == single.scatt_stats()
        assert v.avg_energy() == single.avg_energy()
        oa, ob = v.get_output(), single.get_output()
        assert len(oa["p0"]) == lens[r] - (1 if r == 0 else 0)
        for k in oa:
            assert np.array_equal(oa[k], ob[k]), (r, k)
        assert np.array_equal(v.get_photons_range(0, lens[r]), single.get_photons_range(0, lens[r]))
        s = summ[r]
        mm, ss = v.ph_minmax(), v.scatt_stats()
        assert (s.min_r, s.max_r, s.min_theta, s.max_theta) == mm
        assert (s.max_scatt, s.min_scatt) == ss[:2]
        assert s.avg_scatt == pytest.approx(ss[2], rel=1e-13) and s.avg_r == pytest.approx(ss[3], rel=1e-13)
        assert s.avg_energy == pytest.approx(v.avg_energy(), rel=1e-13)
        assert s.num_output == len(oa["p0"]) and s.list_capacity == lens[r]
        single.close()
    # the pool context itself sees all lists: printPhotons' columns of the lists one after the other
    allout = pool.get_output()
    assert len(allout["p0"]) == sum(lens) - 1
    cat = np.concatenate([pool.pool_rank(r, r).get_output()["r0"] for r in range(3)])
    assert np.array_equal(allout["r0"], cat)
    mm = pool.ph_minmax()
    assert mm[0] == min(s.min_r for s in summ[:3]) and mm[1] == max(s.max_r for s in summ[:3])
    pool.close()


def test_pool_refusals(hip):
    frame, ph, cfg = synth.config1(n_photons=600, n0=8, n1=8)
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(2, 64)                            # slots per rank are rounded up to 512
    v = pool.pool_rank(0, 0)
    with pytest.raises(hip.McratHipError, match="slots per rank"):
        v.set_photons(ph)                              # 600 photons do not fit
    with pytest.raises(hip.McratHipError, match="pool"):
        v.set_hydro(frame)                             # the hydro frame is the pool's
    with pytest.raises(hip.McratHipError, match="views"):
        pool.set_photons(ph)                           # photons go in through the views
    with pytest.raises(hip.McratHipError):
        pool.pool_rank(2, 0)
    v.set_photons(_cut(ph, 0, 400, 600))
    assert v.get_photons()["p0"].shape == (400,)
    pool.pool_create(3, 1000)                          # re-creating drops the lists
    assert pool.pool_summaries()[0].list_capacity == 0
    pool.close()


def test_two_pools_on_two_streams_driven_by_two_host_threads(hip):
    """the adopted ranks are asynchronous (the reference's MPI ranks are processes): a process may deal them out to several pools, each a context
    of its own on its own HIP stream with its own host thread (bench.py --pools) -- every list ends up as in one pool"""
    import threading
    import torch
    n, per, R = 12000, 1000, 12
    frame, ph, cfg = synth.config2(n_photons=n, nzc=8, lumi=3e53)
    rem = 1.0 / frame["fps"]

    def make(lo, hi, stream=None):
        p = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream)
        p.set_hydro(frame)
        p.pool_create(hi - lo, per)
        for r in range(lo, hi):
            p.pool_rank(r - lo, 50 + r).set_photons(_cut(ph, r * per, (r + 1) * per, n))
        p.snapshot_photons()
        return p
    one = make(0, R)
    ref = []
    for k in range(3):                                                     # three frames from the snapshot, the last one is compared
        one.restore_photons()
        one.begin_frame(900 + k, 0.0, rem)
        tot = one.run(0)
    ref = [one.pool_rank(r, 50 + r).get_photons() for r in range(R)]
    assert tot.frame_scatt_cnt > 1000
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    halves = [make(0, 5, streams[0].cuda_stream), make(5, R, streams[1].cuda_stream)]
    halves[1].share_hydro(halves[0])                                       # both pools are in the same hydro frame: one staged copy of it
    stats, errors = [None, None], []

    def drive(p):
        try:
            halves[p].bind_thread()
            for k in range(3):
                halves[p].restore_photons()
                halves[p].begin_frame(900 + k, 0.0, rem)
                stats[p] = halves[p].run(0)
        except Exception as ex:                                            # surfaced below: a thread must not fail silently
            errors.append(ex)
    threads = [threading.Thread(target=drive, args=(p,)) for p in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert stats[0].frame_scatt_cnt + stats[1].frame_scatt_cnt == tot.frame_scatt_cnt
    for p, (lo, hi) in enumerate(((0, 5), (5, R))):
        for r in range(lo, hi):
            out = halves[p].pool_rank(r - lo, 50 + r).get_photons()
            for k in FLOAT_FIELDS + INT_FIELDS:
                assert np.array_equal(out[k], ref[r][k], equal_nan=True), (r, k)
    # the borrower stages a frame of its own again: the arrangement ends, results unchanged
    halves[1].set_hydro(frame)
    halves[1].restore_photons()
    halves[1].begin_frame(902, 0.0, rem)
    again = halves[1].run(0)
    assert again.frame_scatt_cnt == stats[1].frame_scatt_cnt
    other = hip.Engine(synth.TWO, synth.SPHERICAL, cfg["stokes"])
    with pytest.raises(hip.McratHipError, match="GEOMETRY"):
        other.share_hydro(halves[0])
    other.close()
    # The owner stages its next frame while a pool still reads the old one: the reader is cut loose first (its stream drained, no dangling
    # pointers into buffers that are rewritten or freed) and says so until it shares or stages again; then it runs as before.
    halves[1].share_hydro(halves[0])
    halves[0].set_hydro(frame)                                             # (the same cells: the results below must not change)
    halves[1].restore_photons()
    with pytest.raises(hip.McratHipError, match="shared is gone"):
        halves[1].begin_frame(902, 0.0, rem)
    halves[1].share_hydro(halves[0])
    halves[1].restore_photons()
    halves[1].begin_frame(902, 0.0, rem)
    assert halves[1].run(0).frame_scatt_cnt == stats[1].frame_scatt_cnt
    # ... and an owner that is destroyed takes no reader's frame with it silently
    halves[0].close()
    halves[1].restore_photons()
    with pytest.raises(hip.McratHipError):
        halves[1].begin_frame(902, 0.0, rem)
    halves[1].set_hydro(frame)
    halves[1].restore_photons()
    halves[1].begin_frame(902, 0.0, rem)
    assert halves[1].run(0).frame_scatt_cnt == stats[1].frame_scatt_cnt
    for e in [one, halves[1]]:
        e.close()


def test_pool_injection_gives_every_list_the_photons_of_its_own_injection(hip):
    """mcrat_hip_pool_inject_photons (one slab search per group of lists, one workgroup per list) against mcrat_hip_inject_photons on the views"""
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    jet = hip.Engine.outflow(3, lumi=2e53, theta_j=0.1)
    dom = dict(r0_domain=(1e11, 4e12), r1_domain=(0.0, 0.6), r2_domain=(0.0, 0.0))
    fps = 5.0
    specs = [dict(r_inj=1e12, ph_weight=1e50, min_photons=300, max_photons=900, spect="b", theta_min=0.0, theta_max=0.05, seed=11),
             dict(r_inj=1e12, ph_weight=1e50, min_photons=300, max_photons=900, spect="b", theta_min=0.0, theta_max=0.05, seed=12),
             None,
             dict(r_inj=1e12, ph_weight=3e49, min_photons=200, max_photons=600, spect="b", theta_min=0.05, theta_max=0.1, seed=13),
             dict(r_inj=1e12, ph_weight=1e50, min_photons=300, max_photons=900, spect="w", theta_min=0.0, theta_max=0.05, seed=14),
             dict(r_inj=1e12, ph_weight=1e44, min_photons=300, max_photons=900, spect="b", theta_min=0.0, theta_max=0.05, seed=15)]   # the weight loop has to climb
    R = len(specs)
    slab = dict(r_inj=1e12, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=fps, **dom)
    pools = []
    for batched in (True, False):
        pool = hip.Engine(synth.TWO, synth.SPHERICAL, 1)
        pool.ingest(raw, slab, jet)
        pool.pool_create(R, 1024)
        views = [pool.pool_rank(r, 70 + r) for r in range(R)]
        if batched:
            got = pool.pool_inject_photons(fps, specs)
        else:
            got = [None if q is None else views[r].inject_photons(q["r_inj"], q["ph_weight"], q["min_photons"], q["max_photons"], q["spect"], q["theta_min"],
                                                                  q["theta_max"], fps, q["seed"]) for r, q in enumerate(specs)]
        pools.append((pool, views, got))
    (pa, va, ga), (pb, vb, gb) = pools
    assert ga == gb and ga[2] is None and ga[5][1] > 1e44
    assert len({g[0] for g in ga if g}) > 2                                 # Poisson-sized lists
    for r, q in enumerate(specs):
        if q is None:
            assert pa.pool_summaries()[r].list_capacity == 0
            continue
        a, b = va[r].get_photons(), vb[r].get_photons()
        assert len(a["p0"]) == ga[r][0] and q["min_photons"] <= ga[r][0] <= q["max_photons"]
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=(np.asarray(a[k]).dtype.kind == "f")), (r, k)
    # the batched lists run like any others (the window without a list sits the frame out)
    mm = pa.ph_minmax()
    pa.ingest(raw, dict(slab, ph_inj_switch=0, min_r=mm[0], max_r=mm[1], min_theta=0.0, max_theta=0.12), jet)
    pa.begin_frame(5, 1.0, 1.0 / fps)
    pa.run(0)
    for r, q in enumerate(specs):
        if q is not None:
            assert va[r].frame_statistics().remaining_time == 0.0
    pa.close()
    pb.close()


@pytest.mark.parametrize("case", ["2000+5000-cylindrical", "2000+4000-cylindrical-stokes", "3000-spherical-stokes"])
@pytest.mark.parametrize("threads", ["auto", "256"])
def test_lists_of_thousands_of_photons_equal_the_oracle(hip, oracle, case, threads, monkeypatch):
    """the reference's ranks hold 1000 - 5000 photons (sample_mc.par:21-22, Doc/mcrat_doc.tex:165-166): lists of 2000 - 5000 photons, list by
    list against the oracle -- with 512 threads per list (what the engine picks when there are no more such lists than CUs: the columns in
    LDS up to 4096 slots, in HBM/L2 beyond) and with 256"""
    if threads != "auto":
        monkeypatch.setenv("MCRAT_HIP_RANK_BLOCK", threads)
    if case == "2000+5000-cylindrical":
        lens, (frame, ph, cfg) = [2000, 5000], synth.config2(n_photons=7000, nzc=8, stokes=0, lumi=3e52)
    elif case == "2000+4000-cylindrical-stokes":
        lens, (frame, ph, cfg) = [2000, 4000], synth.config2(n_photons=6000, nzc=8, stokes=1, lumi=1e54)
    else:
        lens, (frame, ph, cfg) = [3000, 1100], synth.config3(n_photons=4100, nr=256, nth=128, lumi=1e54)
    subs = _lists(ph, lens)
    R = len(lens)
    seeds, streams = [501 + 13 * r for r in range(R)], [5, 77][:R]
    t0, rem, passes = 1.5, 0.2, 60
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(R, max(lens))
    for r in range(R):
        v = pool.pool_rank(r, streams[r])
        v.set_photons(subs[r])
        v.begin_frame(seeds[r], t0, rem)
    pool.run(passes)
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)   # (bit-identical to the faithful port, tests/test_oracle_kat.py)
    for r in range(R):
        P = oracle.OraclePhotons(synth.photons_to_aos(subs[r], oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seeds[r], time_now=t0, remaining_time=rem, max_iterations=passes, stream=streams[r])
        v = pool.pool_rank(r, streams[r])
        st = v.frame_statistics()
        assert (st.iterations, st.frame_scatt_cnt, st.kn_rejections, st.num_photons_find_new_element) == \
               (rst.iterations, rst.frame_scatt_cnt, rst.kn_rejections, rst.num_photons_find_new_element)
        assert st.last_scattered_index == rst.last_scattered_index
        assert st.time_now == pytest.approx(rtn, rel=1e-12)
        _compare(v.get_photons(), P.aos)
        assert rst.frame_scatt_cnt > 0
    pool.close()


def test_pool_set_photons_equals_set_photons_list_by_list(hip):
    """mcrat_hip_pool_set_photons (a CONTINUE run's restart: all adopted ranks' checkpointed lists in one copy and one launch) leaves every list
    exactly as mcrat_hip_set_photons on its view does -- columns, derived columns and flags (a frame run afterwards is bit-identical), the rest of
    the window cleared, lists that were not named untouched; bad requests are refused before anything is touched"""
    frame, ph, cfg = synth.config2(n_photons=4000, nzc=8, stokes=1, lumi=1e54)
    lens = [700, 1000, 3, 1024, 512]
    subs = _lists(ph, lens)
    recs = [synth.photons_to_aos(sp, hip.PHOTON_DTYPE) for sp in subs]
    rem = 1.0 / frame["fps"]

    def pool():
        p = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
        p.set_hydro(frame)
        p.pool_create(len(lens) + 1, 1100)
        return p
    one = pool()
    for r in range(len(lens)):
        one.pool_rank(r, 20 + r).set_photons_aos(recs[r])
    one.begin_frame(4, 0.0, rem)
    want_stats = one.run(80)
    want = [one.pool_rank(r, 20 + r).get_photons() for r in range(len(lens))]
    many = pool()
    for r in range(len(lens) + 1):
        many.pool_rank(r, 20 + r)
    many.pool_rank(5, 25).set_photons_aos(recs[2])                      # a list that is not named below keeps its photons
    many.pool_rank(1, 21).set_photons_aos(recs[3])                      # a longer list was here before: its tail must be cleared
    with pytest.raises(hip.McratHipError):
        many.pool_set_photons([0, 0], [recs[0], recs[1]])               # a rank twice
    with pytest.raises(hip.McratHipError):
        many.pool_set_photons([0, 9], [recs[0], recs[1]])               # no such rank
    many.pool_set_photons([3, 0, 1, 2, 4], [recs[3], recs[0], recs[1], recs[2], recs[4]])
    assert [many.pool_rank(r, 20 + r).n for r in range(6)] == lens + [lens[2]]
    kept = many.pool_rank(5, 25).get_photons()
    assert np.array_equal(kept["p0"], subs[2]["p0"])
    many.begin_frame(4, 0.0, rem)
    got_stats = many.run(80)
    for r in range(len(lens)):
        out = many.pool_rank(r, 20 + r).get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(out[k], want[r][k], equal_nan=True), (r, k)
    assert got_stats.frame_scatt_cnt >= want_stats.frame_scatt_cnt      # (the extra list scatters too)
    one.close()
    many.close()


def test_profile_totals_count_the_loop_kernels_launches(hip):
    """mcrat_hip_profile_totals: the summed duration and number of the loop kernel's launches of a profile = 1 context (bench.py's loop-only roofline)"""
    frame, ph, cfg = synth.config2(n_photons=4000, nzc=8, lumi=1e53)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=1000, profile=True)
    e.set_hydro(frame)
    e.set_photons(ph)
    assert e.profile_totals() == (0.0, 0)
    e.begin_frame(5, 0.0, 1.0 / frame["fps"])
    st = e.run(0)
    ms, launches = e.profile_totals()
    assert launches >= 1 and 0 < ms < 1e4 and st.step_kernel_launches == launches and abs(st.step_kernel_ms - ms) < 1e-9
    e.close()
    plain = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    with pytest.raises(hip.McratHipError):
        plain.profile_totals()
    plain.close()
