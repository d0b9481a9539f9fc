"""GPU parity tests: the HIP path (through the C ABI, libmcrat_hip.so) against the CPU oracle
on the same seeded inputs.

Bar (BASELINE.json north_star): integers -- cell indices, scattering
counts, which photon scattered in which iteration -- are exact; doubles agree to a relative
1e-9 over whole trajectories (the two sides share the algorithm, the random stream and
-ffp-contract=off arithmetic and differ only in libm's last-ulp rounding of
log/atan2/sincos/acos/exp; boosting in and out of a Gamma = 100 flow amplifies an ulp by
~Gamma^2, and the trajectory compounds it).  4-vector components are compared relative to the
vector's norm.  The headline requirement is 1e-5 on 4-momenta and Stokes parameters.
"""
import ctypes as C

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-9
FLOAT_FIELDS = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2",
                "s0", "s1", "s2", "s3", "total_optical_depth", "time_to_scatter")
INT_FIELDS = ("nearest_block_index", "num_scatt", "recalc_properties", "weight")


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _oracle_run(oracle, frame, ph, cfg, seed, time_now, remaining, max_iterations=0):
    H = oracle.OracleHydro(frame)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    st, tn, rem, _ = oracle.photon_loop(c, P, H, seed=seed, time_now=time_now, remaining_time=remaining,
                                        max_iterations=max_iterations)
    return P.aos, st, tn, rem


def _gpu_run(hip, frame, ph, cfg, seed, time_now, remaining, max_iterations=0, **kw):
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], **kw)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(seed, time_now, remaining)
    st = e.run(max_iterations)
    out = e.get_photons()
    return e, out, st


def _compare(gpu, ref, rtol=RTOL):
    for k in INT_FIELDS:
        assert np.array_equal(np.asarray(gpu[k]), ref[k]), k
    assert np.array_equal(np.asarray(gpu["type"]), ref["type"])
    for k in FLOAT_FIELDS:
        a, b = np.asarray(gpu[k]), ref[k]
        scale = np.maximum(np.abs(b), 1e-300)
        if k in ("p1", "p2", "p3"):
            scale = np.maximum(np.abs(ref["p0"]), 1e-300)   # vector components: error relative to |p| = p0
        if k in ("comv_p1", "comv_p2", "comv_p3"):
            scale = np.maximum(np.abs(ref["comv_p0"]), 1e-300)
        if k in ("r0", "r1", "r2"):
            scale = np.maximum(scale, 1e9)
        if k in ("s0", "s1", "s2", "s3"):
            scale = np.ones_like(b)                   # Stokes parameters are fractions of I = 1: absolute error
        err = np.abs(a - b) / scale
        assert np.all(err <= rtol), (k, float(err.max()), int(err.argmax()))


# ------------------------------------------------------------------ cell search
@pytest.mark.parametrize("which", ["flash", "pluto", "cart3d", "sph3d", "polar3d"])
def test_cell_lookup_equals_linear_scan(hip, oracle, which):
    """device bucket search == the reference's lowest-index linear scan (geometry.c:350-391), including
    points exactly on shared faces and corners (closed intervals -> two or four containing cells)."""
    rng = np.random.default_rng(1)
    if which == "flash":
        frame, _, cfg = synth.config2(n_photons=64, nzc=4)
    elif which == "pluto":
        frame, _, cfg = synth.config3(n_photons=64, nr=96, nth=48)
    elif which == "cart3d":
        frame, _, cfg = synth.config_3d_cartesian(n_photons=64, n=(8, 8, 8))
    elif which == "sph3d":
        frame, _, cfg = synth.config_3d(synth.SPHERICAL, n_photons=64)
    else:
        frame, _, cfg = synth.config_3d(synth.POLAR, n_photons=64)
    three = cfg["dimensions"] == synth.THREE
    M = frame["num_elements"]
    pick = rng.integers(0, M, 1500)
    u = rng.random((3, pick.size)) - 0.5
    # every 3rd point is snapped to a face / corner of its cell
    snap = (np.arange(pick.size) % 3 == 0)
    u[:, snap] = np.sign(u[:, snap]) * 0.5
    u[1, snap & (np.arange(pick.size) % 2 == 0)] = 0.123
    a0 = frame["r0"][pick] + u[0] * frame["r0_size"][pick]
    a1 = frame["r1"][pick] + u[1] * frame["r1_size"][pick]
    a2 = frame["r2"][pick] + u[2] * frame["r2_size"][pick] if three else np.zeros(pick.size)
    # plus points outside the mesh
    a0 = np.concatenate([a0, [frame["r0"].max() * 3.0, -1.0 if three else frame["r0"].max() * 1.5]])
    a1 = np.concatenate([a1, [frame["r1"].max() * 3.0, frame["r1"].mean()]])
    a2 = np.concatenate([a2, [0.0, 0.0]])

    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    e.set_hydro(frame)
    got = e.lookup_cell(a0, a1, a2 if three else None)

    L = oracle.lib()
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], 0)
    want = np.array([L.orc_findContainingBlock(C.byref(c), float(x), float(y), float(z), C.byref(H.c))
                     for x, y, z in zip(a0, a1, a2)], dtype=np.int32)
    assert np.array_equal(got, want)
    assert np.array_equal(want[:-2][~snap], pick[~snap])            # interior points: the cell they were drawn in
    # a face point belongs to the lowest-index cell touching it, or to none when centre +- size/2 rounds
    # one ulp past the neighbour's face (both sides apply the same closed-interval arithmetic)
    assert (want[:-2][snap] != pick[snap]).sum() > 10
    assert want[-2] == -1


# ------------------------------------------------------------------ first half of an iteration
@pytest.mark.parametrize("make", ["cfg1", "cfg2", "cfg3"])
def test_locate_and_sample_step(hip, oracle, make):
    """findContainingHydroCell (forced) + calcMeanFreePath: cell index exact; comoving momentum, optical
    depth and sampled free time to 1e-10 (one boost into a Gamma=100 frame costs ~Gamma^2 ulp)."""
    if make == "cfg1":
        frame, ph, cfg = synth.config1(n_photons=3000, n0=32, n1=32)
    elif make == "cfg2":
        frame, ph, cfg = synth.config2(n_photons=3000, nzc=8)
    else:
        frame, ph, cfg = synth.config3(n_photons=3000, nr=256, nth=128)
    seed = 2024
    L = oracle.lib()
    H = oracle.OracleHydro(frame)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    rng = oracle.Rng()
    L.orc_rng_init(C.byref(rng), seed, 0)
    L.orc_rng_set_iteration(C.byref(rng), 0)
    st = oracle.Stats()
    L.orc_findContainingHydroCell(C.byref(c), C.byref(P.c), C.byref(H.c), 1, C.byref(st))
    L.orc_calcMeanFreePath(C.byref(c), C.byref(P.c), C.byref(H.c), C.byref(rng))

    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(seed, 0.0, 0.2)
    e.step_locate_sample(1)
    out = e.get_photons()
    assert (P.aos["nearest_block_index"] >= 0).all()
    _compare(out, P.aos, rtol=1e-10)
    # the event half walks the candidates exactly as the oracle's photonEvent does on its sorted list: the same photon reported,
    # the same time step, the same number of Klein-Nishina rejections on the way, and the same photons afterwards
    s = e.step_event()
    idx, scatt = C.c_int(-1), C.c_longlong(0)
    ost = oracle.Stats()
    o_dt = L.orc_photonEvent(C.byref(c), C.byref(P.c), 0.2, C.byref(H.c), C.byref(idx), C.byref(scatt), C.byref(rng), C.byref(ost))
    assert s.iterations == 1
    assert s.last_scattered_index == idx.value
    assert s.frame_scatt_cnt == scatt.value
    assert s.kn_rejections == ost.kn_rejections
    assert s.last_time_step == pytest.approx(o_dt, rel=1e-12)
    if ost.kn_rejections == 0 and scatt.value == 1:
        assert idx.value == P.sorted[0]                  # the head of the argsort (mclib.c:702-712) scattered
    _compare(e.get_photons(), P.aos, rtol=1e-9)          # (the read-back applies the advance the walk left pending)


# ------------------------------------------------------------------ trajectories
CASES = {
    # name: (factory, kwargs, iterations)
    "cfg1-cartesian-wind": (synth.config1, dict(n_photons=2000, n0=32, n1=32), 1500),
    # the jet of the BASELINE configs is optically thin at the injection radius (a few scatterings per
    # thousand photons per frame); a brighter jet makes 10^3 events fit in the test
    "cfg2-cylindrical-jet": (synth.config2, dict(n_photons=2000, nzc=8, lumi=1e54), 1200),
    "cfg2-cylindrical-jet-stokes": (synth.config2, dict(n_photons=1500, nzc=8, stokes=1, lumi=1e54), 1000),
    "cfg3-spherical-jet-stokes": (synth.config3, dict(n_photons=2000, nr=256, nth=128, lumi=1e54), 1000),
    "cfg2-cylindrical-jet-thin": (synth.config2, dict(n_photons=2000, nzc=8), 0),
    # ten times closer to the engine the jet is at 3e7 K: the Maxwell-Juttner branch (electron.c:207-226), whose rejection loop takes
    # ~150 attempts per electron -- several 64-attempt rounds of the event walk's wavefront (physics.hpp, sample_thermal_electron)
    "cfg2-hot-inner-jet-stokes": (synth.config2, dict(n_photons=2000, nzc=8, stokes=1, lumi=1e54, r_inj=1e11, block_side=2.5e7), 1000),
    # just above the 1e7 K switch (1.7e7 K) the sampler accepts one attempt in ~250: four or five rounds per electron
    "cfg2-warm-inner-jet": (synth.config2, dict(n_photons=2000, nzc=8, lumi=1e53, r_inj=1e11, block_side=2.5e7), 800),
    "3d-cartesian-wind-stokes": (synth.config_3d_cartesian, dict(n_photons=1500), 800),
    # every (DIMENSIONS, GEOMETRY) pair the reference supports (mcrat.h:196-204)
    "2.5d-cylindrical-toroidal-flow-stokes": (synth.config_25d, dict(geometry=synth.CYLINDRICAL), 700),
    "2.5d-spherical-toroidal-flow-stokes": (synth.config_25d, dict(geometry=synth.SPHERICAL), 700),
    "3d-spherical-wind-stokes": (synth.config_3d, dict(geometry=synth.SPHERICAL), 700),
    "3d-polar-wind-stokes": (synth.config_3d, dict(geometry=synth.POLAR), 700),
    "2d-cartesian-wind-stokes": (synth.config1, dict(n_photons=1500, n0=32, n1=32, stokes=1), 700),
}


@pytest.mark.parametrize("name", list(CASES))
def test_trajectory_parity(hip, oracle, name):
    factory, kw, iters = CASES[name]
    frame, ph, cfg = factory(**kw)
    seed, t0, rem = 0x4D435261, 3.0, 1.0 / frame["fps"]
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, t0, rem, iters)
    e, out, st = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters)
    assert st.iterations == rst.iterations and (iters == 0 or st.iterations == iters)
    assert st.frame_scatt_cnt == rst.frame_scatt_cnt > 0
    assert st.kn_rejections == rst.kn_rejections
    assert st.num_photons_find_new_element == rst.num_photons_find_new_element
    assert st.not_found == rst.not_found
    assert st.last_scattered_index == rst.last_scattered_index
    assert st.time_now == pytest.approx(rtn, rel=1e-12) and st.remaining_time == pytest.approx(rrem, rel=1e-9)
    _compare(out, ref)
    if cfg["stokes"]:
        pol = np.hypot(out["s1"], out["s2"])
        assert pol.max() > 0.01 and (pol <= 1 + 1e-9).all()


def test_hot_plasma_maxwell_juttner_and_kn_rejection_chains(hip, oracle):
    """T >= 1e7 K takes the Maxwell-Juttner branch (electron.c:207-226, needs K_2) and MeV photons are
    Klein-Nishina rejected often, so the candidate walk of photonEvent (mclib.c:1128-1339) goes several
    photons deep and the candidate list has to be refilled."""
    frame, ph, cfg = synth.config1(n_photons=600, n0=16, n1=16)
    frame["temp"] = np.full(frame["num_elements"], 4e9)
    for k in ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
        ph[k] = ph[k] * 300.0            # ~ MeV in the fluid frame -> sigma_KN/sigma_T well below 1
    seed, iters = 77, 600
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, 0.0, 0.2, iters)
    e, out, st = _gpu_run(hip, frame, ph, cfg, seed, 0.0, 0.2, iters)
    assert rst.kn_rejections > 200
    assert st.kn_rejections == rst.kn_rejections and st.frame_scatt_cnt == rst.frame_scatt_cnt
    assert st.rescans > 0
    _compare(out, ref)


def test_whole_frame_and_photons_leaving_the_domain(hip, oracle):
    """run a (short) frame to completion: the loop ends exactly at the frame time, photons that leave the
    hydro domain get index -1, the 1e12/c free time, and never scatter (mclib.c:589-593,682-687)."""
    frame, ph, cfg = synth.config1(n_photons=500, n0=16, n1=16)
    frame["r1_domain"] = (0.0, 1e12 + 2.0e9)        # the domain ends just above the injection shell
    seed, t0, rem = 5, 0.0, 0.1
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, t0, rem)
    e, out, st = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iterations_per_sync=64)
    assert rrem == 0.0 and st.remaining_time == 0.0
    assert st.iterations == rst.iterations and st.frame_scatt_cnt == rst.frame_scatt_cnt
    assert st.time_now == pytest.approx(t0 + rem, rel=1e-13)
    gone = ref["nearest_block_index"] == -1
    assert gone.sum() > 20 and (~gone).sum() > 20
    assert (ref["time_to_scatter"][gone] == 1e12 / synth.C_LIGHT).all()
    _compare(out, ref)
    # propagate_frame is begin_frame + run
    e2 = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e2.set_hydro(frame)
    e2.set_photons(ph)
    tn, st2 = e2.propagate_frame(t0, rem, seed)
    out2 = e2.get_photons()
    assert tn == st.time_now and st2.iterations == st.iterations
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(out2[k], out[k]), k


def test_split_runs_graph_and_profile_modes_are_bitwise_identical(hip):
    frame, ph, cfg = synth.config2(n_photons=3000, nzc=8, stokes=1, lumi=1e54)
    seed, t0, rem, iters = 11, 0.0, 0.2, 300
    _, base, st0 = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters)
    variants = {
        "graph": dict(use_graph=True, iterations_per_sync=50),
        "profile": dict(profile=True, iterations_per_sync=64),
        "small-batches": dict(iterations_per_sync=7),
    }
    for name, kw in variants.items():
        _, out, st = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters, **kw)
        assert st.iterations == iters and st.frame_scatt_cnt == st0.frame_scatt_cnt, name
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(out[k], base[k]), (name, k)
        if name == "profile":
            assert st.step_kernel_launches == iters and st.step_kernel_ms > 0 and st.event_kernel_ms > 0
    # a captured graph carries the frame's key: a second frame on the same context with another seed must not replay the first one's
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], use_graph=True, iterations_per_sync=50)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.snapshot_photons()
    e.begin_frame(seed + 1, t0, rem)
    e.run(iters)
    other = e.get_photons()
    e.restore_photons()
    e.begin_frame(seed, t0, rem)
    e.run(iters)
    again = e.get_photons()
    e.close()
    assert not np.array_equal(other["p0"], base["p0"])
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(again[k], base[k]), ("graph after a seed change", k)
    # the same frame in three calls
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(seed, t0, rem)
    total = 0
    for n in (100, 1, 199):
        mid = e.get_photons()          # reading photons back mid-frame must not disturb the run
        total = e.run(n).iterations
    out = e.get_photons()
    assert total == iters
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(out[k], base[k]), k


def test_different_seed_and_rank_stream_change_the_result(hip):
    frame, ph, cfg = synth.config1(n_photons=1000, n0=16, n1=16)
    _, a, _ = _gpu_run(hip, frame, ph, cfg, 1, 0.0, 0.2, 200)
    _, b, _ = _gpu_run(hip, frame, ph, cfg, 2, 0.0, 0.2, 200)
    _, c, _ = _gpu_run(hip, frame, ph, cfg, 1, 0.0, 0.2, 200, rng_stream=3)
    assert not np.array_equal(a["p0"], b["p0"]) and not np.array_equal(a["p0"], c["p0"])


# ------------------------------------------------------------------ boundary marshalling
def test_aos_round_trip_and_null_photon_slots(hip, oracle):
    """struct photon records (176 B, NULL slots of photons.c:208 included) survive H->D->H unchanged,
    and NULL / zero-weight / CS-pool photons are handled as the reference does."""
    frame, ph, cfg = synth.config1(n_photons=777, n0=16, n1=16)
    aos = synth.photons_to_aos(ph, hip.PHOTON_DTYPE)
    null = np.arange(5, 777, 50)
    for k in aos.dtype.names:
        if k != "type":
            aos[k][null] = 0
    aos["type"][null] = b"N"
    aos["nearest_block_index"][null] = -1
    aos["type"][7] = b"p"                 # a cyclo-synchrotron pool photon does not stream (mclib.c:1070)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    e.set_hydro(frame)
    e.set_photons_aos(aos)
    back = e.get_photons_aos()
    for k in aos.dtype.names:
        assert np.array_equal(back[k], aos[k]), k
    e.begin_frame(9, 0.0, 0.2)
    st = e.run(300)
    got = e.get_photons_aos()
    H = oracle.OracleHydro(frame)
    P = oracle.OraclePhotons(aos)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], 0)
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=9, time_now=0.0, remaining_time=0.2, max_iterations=300)
    assert st.frame_scatt_cnt == rst.frame_scatt_cnt
    _compare({k: got[k] for k in got.dtype.names}, P.aos)
    assert (got["r0"][null] == 0).all() and (got["num_scatt"][null] == 0).all()
    assert got["r0"][7] == aos["r0"][7] and got["r2"][7] == aos["r2"][7]


def test_per_frame_reductions(hip, oracle):
    L = oracle.lib()
    frame, ph, cfg = synth.config2(n_photons=5000, nzc=8, lumi=1e54)
    e, out, st = _gpu_run(hip, frame, ph, cfg, 3, 0.0, 0.2, 200)
    P = oracle.OraclePhotons(synth.photons_to_aos(out, oracle.PHOTON_DTYPE))
    a, b, c_, d = (C.c_double() for _ in range(4))
    L.orc_phMinMax(C.byref(P.c), C.byref(a), C.byref(b), C.byref(c_), C.byref(d))
    got = e.ph_minmax()
    assert got[0] == a.value and got[1] == b.value
    assert got[2] == pytest.approx(c_.value, rel=1e-12) and got[3] == pytest.approx(d.value, rel=1e-12)
    mx, mn = C.c_int(), C.c_int()
    avg, ravg = C.c_double(), C.c_double()
    L.orc_phScattStats(C.byref(P.c), C.byref(mx), C.byref(mn), C.byref(avg), C.byref(ravg))
    g = e.scatt_stats()
    assert (g[0], g[1]) == (mx.value, mn.value)
    assert g[2] == pytest.approx(avg.value, rel=1e-12) and g[3] == pytest.approx(ravg.value, rel=1e-12)
    assert e.avg_energy() == pytest.approx(L.orc_averagePhotonEnergy(C.byref(P.c)), rel=1e-12)


def test_errors_are_reported_not_fatal(hip):
    with pytest.raises(hip.McratHipError):
        hip.Engine(synth.TWO, synth.POLAR)                 # unsupported (DIMENSIONS, GEOMETRY) pair
    e = hip.Engine(synth.TWO, synth.CYLINDRICAL)
    with pytest.raises(hip.McratHipError):
        e.begin_frame(1, 0.0, 0.2)                         # no hydro / photons staged yet
    with pytest.raises(hip.McratHipError):
        e.run(1)


# ------------------------------------------------------------------ virtual ranks
@pytest.mark.parametrize("case", ["cfg1", "cfg2-hot", "cfg2-stokes", "cfg3-stokes"])
def test_virtual_ranks_equal_independent_lists(hip, oracle, case):
    """virtual_rank_photons = n: every block of n consecutive slots is an independent photon list with its own
    clock and RNG stream rng_stream + r -- the reference's many-ranks run shape.  Each must equal the oracle run
    on that sub-list alone (slot indices restart at 0, stream = first_stream + r), including the short last rank."""
    if case == "cfg1":
        frame, ph, cfg = synth.config1(n_photons=3000, n0=32, n1=32)
    elif case == "cfg2-hot":
        frame, ph, cfg = synth.config2(n_photons=3000, nzc=8, lumi=1e54, r_inj=1e11, block_side=2.5e7)
        assert frame["temp"].min() > 1e7
    elif case == "cfg2-stokes":
        frame, ph, cfg = synth.config2(n_photons=3000, nzc=8, stokes=1, lumi=1e54)
    else:
        frame, ph, cfg = synth.config3(n_photons=3000, nr=256, nth=128, lumi=1e54)
    per, first_stream, seed, t0, rem, passes = 700, 5, 31, 2.0, 0.2, 300
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=first_stream, virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    assert e.num_virtual_ranks() == 5
    e.begin_frame(seed, t0, rem)
    tot = e.run(passes)
    out = e.get_photons()
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    n = 3000
    sums = dict(it=0, sc=0, rel=0, rej=0, steps=0)
    for r in range(5):
        lo, hi = r * per, min((r + 1) * per, n)
        sub = {k: (v[lo:hi].copy() if isinstance(v, np.ndarray) else v) for k, v in ph.items()}
        P = oracle.OraclePhotons(synth.photons_to_aos(sub, oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seed, time_now=t0, remaining_time=rem, max_iterations=passes,
                                               stream=first_stream + r)
        st = e.rank_stats(r)
        assert st.iterations == rst.iterations == passes
        assert st.frame_scatt_cnt == rst.frame_scatt_cnt > 0
        assert st.kn_rejections == rst.kn_rejections
        assert st.num_photons_find_new_element == rst.num_photons_find_new_element
        assert st.time_now == pytest.approx(rtn, rel=1e-12) and st.remaining_time == pytest.approx(rrem, rel=1e-9)
        assert st.last_scattered_index - lo == rst.last_scattered_index
        _compare({k: v[lo:hi] for k, v in out.items()}, P.aos)
        sums["it"] += rst.iterations; sums["sc"] += rst.frame_scatt_cnt; sums["rel"] += rst.num_photons_find_new_element
        sums["rej"] += rst.kn_rejections; sums["steps"] += rst.iterations * (hi - lo)
    got = (tot.iterations, tot.frame_scatt_cnt, tot.num_photons_find_new_element, tot.kn_rejections, tot.photon_steps)
    assert got == (sums["it"], sums["sc"], sums["rel"], sums["rej"], sums["steps"])


def test_virtual_ranks_whole_frame_and_split_runs(hip, oracle):
    frame, ph, cfg = synth.config1(n_photons=1500, n0=16, n1=16)
    frame["r1_domain"] = (0.0, 1e12 + 2.0e9)
    per, seed, t0, rem = 500, 8, 0.0, 0.02
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e.set_hydro(frame)
    e.set_photons(ph)
    tn, tot = e.propagate_frame(t0, rem, seed)
    out = e.get_photons()
    assert tn == pytest.approx(t0 + rem, rel=1e-13) and tot.remaining_time == 0.0
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    total = 0
    for r in range(3):
        sub = {k: (v[r * per:(r + 1) * per].copy() if isinstance(v, np.ndarray) else v) for k, v in ph.items()}
        P = oracle.OraclePhotons(synth.photons_to_aos(sub, oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seed, time_now=t0, remaining_time=rem, stream=r)
        assert rrem == 0.0 and e.rank_stats(r).iterations == rst.iterations
        _compare({k: v[r * per:(r + 1) * per] for k, v in out.items()}, P.aos)
        total += rst.frame_scatt_cnt
    assert tot.frame_scatt_cnt == total > 100
    # the same frame in bounded pieces is bitwise the same
    e2 = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
    e2.set_hydro(frame)
    e2.set_photons(ph)
    e2.begin_frame(seed, t0, rem)
    for _ in range(1000):
        mid = e2.get_photons()
        if e2.run(37).remaining_time == 0.0:
            break
    out2 = e2.get_photons()
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(out2[k], out[k]), k


def test_virtual_ranks_are_deterministic_and_equal_single_list_contexts_at_scale(hip, monkeypatch):
    """300 lists on the 1 048 576-cell frame, run three times (once split after the forced pass, once with 128-thread
    workgroups): identical counters and
    photons both times, and a sample of lists bit-identical to single-list contexts holding only their photons
    (rng_stream = first stream + r).  Guards the workgroup-level hand-offs of rank_loop_kernel against races."""
    n, per = 300000, 1000
    frame, ph, cfg = synth.config2(n_photons=n)
    rem = 1.0 / frame["fps"]
    runs = []
    # 128: four lists per CU, same arithmetic; the last run keeps the lists' columns in HBM/L2 instead of LDS (the path long lists take)
    for split, block, no_lds in ((False, "256", False), (True, "256", False), (False, "128", False), (False, "256", True)):
        monkeypatch.setenv("MCRAT_HIP_RANK_BLOCK", block)
        if no_lds:
            monkeypatch.setenv("MCRAT_HIP_NO_LDS_LISTS", "1")
        else:
            monkeypatch.delenv("MCRAT_HIP_NO_LDS_LISTS", raising=False)
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=per)
        e.set_hydro(frame)
        e.set_photons(ph)
        e.begin_frame(1, 0.0, rem)
        if split:
            e.run(1)
        st = e.run(0)
        out = e.get_photons()
        per_rank = [(e.rank_stats(r).iterations, e.rank_stats(r).frame_scatt_cnt) for r in range(0, e.num_virtual_ranks(), 7)]
        runs.append((st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element, per_rank, out))
        e.close()
    for other in runs[1:]:
        assert runs[0][:4] == other[:4]
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(np.asarray(runs[0][4][k]), np.asarray(other[4][k]), equal_nan=True), k
    out = runs[0][4]
    for r in (0, 113, 276, 299):
        lo, hi = r * per, (r + 1) * per
        sub = {k: (v[lo:hi].copy() if hasattr(v, "__len__") and len(v) == n else v) for k, v in ph.items()}
        s = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], rng_stream=r)
        s.set_hydro(frame)
        s.set_photons(sub)
        s.begin_frame(1, 0.0, rem)
        s.run(0)
        o = s.get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(np.asarray(o[k]), np.asarray(out[k])[lo:hi], equal_nan=True), (r, k)
        s.close()


# ------------------------------------------------------------------ TAU_CALCULATION == TABLE
def _hot_table():
    """a smooth stand-in for thermal_hot_x_section.dat on the reference's grid (hot_x_section.h:2-10); the real table
    is created by MCRaT itself (GSL Monte-Carlo integration) and handed to the engine"""
    i, j = np.meshgrid(np.arange(221), np.arange(81), indexing="ij")
    x = -12.0 + i * (18.0 / 220)
    y = -4.0 + j * (8.0 / 80)
    return -0.35 * np.log1p(np.exp(2.0 * (x + 0.5))) / np.log(10) - 0.02 * (y + 4.0) * (1 + 0.1 * np.tanh(x))


@pytest.mark.parametrize("case", ["cfg2-stokes", "cfg1-hot-kn", "cfg2-virtual-ranks"])
def test_table_optical_depth_trajectories(hip, oracle, case):
    """the optical depth carries the interpolated thermal cross section of (comv_p0, T_cell) wherever the reference
    recomputes it: on re-location (mclib.c:570) and after a scatter (mclib.c:668-673)"""
    tab = _hot_table()
    if case == "cfg1-hot-kn":
        frame, ph, cfg = synth.config1(n_photons=600, n0=16, n1=16)
        frame["temp"] = np.full(frame["num_elements"], 4e9)
        for k in ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
            ph[k] = ph[k] * 300.0
        seed, t0, rem, iters = 77, 0.0, 0.2, 500
    else:
        frame, ph, cfg = synth.config2(n_photons=2000, nzc=8, stokes=1, lumi=1e54)
        seed, t0, rem, iters = 0x4D435261, 3.0, 1.0 / frame["fps"], 800
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], hot_table=tab)
    per = 500 if case == "cfg2-virtual-ranks" else 0
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE, virtual_rank_photons=per)
    e.set_hot_cross_section(tab)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(seed, t0, rem)
    st = e.run(iters)
    out = e.get_photons()
    H = oracle.OracleHydro(frame)
    if per == 0:
        P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seed, time_now=t0, remaining_time=rem, max_iterations=iters)
        assert st.iterations == rst.iterations == iters
        assert st.frame_scatt_cnt == rst.frame_scatt_cnt > 100 and st.kn_rejections == rst.kn_rejections
        assert st.table_fallbacks == rst.table_fallbacks == 0
        _compare(out, P.aos)
        # ... and the table matters: the DIRECT trajectory is a different one
        d, dout, dst = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters)
        assert not np.array_equal(dout["num_scatt"], out["num_scatt"])
    else:
        n = len(ph["p0"])
        for r in range(e.num_virtual_ranks()):
            lo, hi = r * per, min(n, (r + 1) * per)
            sub = {k: (v[lo:hi].copy() if hasattr(v, "__len__") and len(v) == n else v) for k, v in ph.items()}
            P = oracle.OraclePhotons(synth.photons_to_aos(sub, oracle.PHOTON_DTYPE))
            rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seed, time_now=t0, remaining_time=rem, max_iterations=iters, stream=r)
            rs = e.rank_stats(r)
            assert rs.iterations == rst.iterations and rs.frame_scatt_cnt == rst.frame_scatt_cnt
            _compare({k: np.asarray(v)[lo:hi] for k, v in out.items()}, P.aos)


def test_table_lookups_outside_the_table_are_integrated_afresh(hip, oracle):
    """hot_x_section.c:563-599: a look-up that falls off the table (GSL_EDOM) integrates the cross section at that (energy, temperature) --
    calculateTotalThermalCrossSection, :324-356 -- and the loop goes on with the value.  A table that starts above the photons' energies: EVERY
    look-up takes the integral (a coarse one here: 2560 samples), in list mode by the lane that meets it.  Values against the oracle, not counts."""
    tab = _hot_table()
    frame, ph, cfg = synth.config1(n_photons=300, n0=16, n1=16)
    grid = (-2.0, 6.0, -4.0, 4.0)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], hot_table=tab, grid=grid, fallback_calls=2560)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
    e.set_hot_cross_section(tab, grid)
    assert e.table_fallback_calls() == 500000 and e.table_fallback_calls(2560) == 2560
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(9, 0.0, 0.2)
    st = e.run(100)
    out = e.get_photons()
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    H = oracle.OracleHydro(frame)
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=9, time_now=0.0, remaining_time=0.2, max_iterations=100)
    # counted per evaluation: the engine evaluates the optical depth of a scattered photon at once (and again if the
    # photon has left its cell by the next pass), the reference at the next pass -- at most one more per scattering
    assert rst.table_fallbacks > 300 and 0 <= st.table_fallbacks - rst.table_fallbacks <= st.frame_scatt_cnt
    assert st.frame_scatt_cnt == rst.frame_scatt_cnt > 0
    # 1e-6, not 1e-9, as in the cold-plasma test below: the integrand holds the reference's Klein-Nishina formula, which just above its 1e-3 seam cancels
    # terms of 2/e^2 ~ 1e6 down to ~1 -- one ulp of log() is 1e-10 of a sample, and the device's log is not glibc's.  Integers stay exact.
    _compare(out, P.aos, rtol=1e-6)


@pytest.mark.parametrize("calls", [2560, 500000])
def test_table_lookups_outside_the_table_in_a_rank_pool(hip, oracle, calls):
    """the same in rank_loop_kernel, which brings every such look-up to a whole wavefront (64 substreams of the integral at a time; the walk's
    wavefront for a scattered photon's new optical depth): the value belongs to (pass, slot, list), so it is the number the oracle's loop computes
    in one thread -- at the reference's 500 000 samples too"""
    from tests.test_gpu_pool import _lists
    tab = _hot_table()
    lens = [70, 130, 64] if calls < 10000 else [24, 40]
    passes = 40 if calls < 10000 else 4
    frame, ph, cfg = synth.config2(n_photons=sum(lens), nzc=8, stokes=1, lumi=1e54)
    subs = _lists(ph, lens)
    grid = (-2.0, 6.0, -4.0, 4.0)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], hot_table=tab, grid=grid, fallback_calls=calls, optimised=True)
    H = oracle.OracleHydro(frame)
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
    pool.set_hot_cross_section(tab, grid)
    pool.table_fallback_calls(calls)
    pool.set_hydro(frame)
    pool.pool_create(len(lens), 256)
    t0, rem = 1.5, 1.0 / frame["fps"]
    for r in range(len(lens)):
        v = pool.pool_rank(r, 3 + r)
        v.set_photons(subs[r])
        v.begin_frame(500 + r, t0, rem)
    pool.run(passes)
    fallbacks = 0
    for r in range(len(lens)):
        P = oracle.OraclePhotons(synth.photons_to_aos(subs[r], oracle.PHOTON_DTYPE))
        rst, rtn, _, _ = oracle.photon_loop(c, P, H, seed=500 + r, time_now=t0, remaining_time=rem, max_iterations=passes, stream=3 + r)
        v = pool.pool_rank(r, 3 + r)
        st = v.frame_statistics()
        assert (st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element) == (rst.iterations, rst.frame_scatt_cnt, rst.num_photons_find_new_element)
        assert st.time_now == pytest.approx(rtn, rel=1e-12)
        _compare(v.get_photons(), P.aos, rtol=1e-6)         # (the integrand's cancellation, see above)
        fallbacks += rst.table_fallbacks
    assert fallbacks >= sum(lens)
    pool.close()


def test_table_lookups_in_cold_plasma_are_the_klein_nishina_cross_section(hip, oracle):
    """cells colder than the table's lowest temperature (5.9e5 K with the reference's LOG_T_MIN): calculateTotalThermalCrossSection returns the
    Klein-Nishina cross section of the photon's comoving energy there (hot_x_section.c:337-340) -- engine and oracle do the same, nothing is clamped"""
    tab = _hot_table()
    frame, ph, cfg = synth.config1(n_photons=300, n0=16, n1=16)
    frame = dict(frame, temp=np.full_like(frame["temp"], 2.0e5))
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], hot_table=tab)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
    e.set_hot_cross_section(tab)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(9, 0.0, 0.2)
    st = e.run(150)
    out = e.get_photons()
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    H = oracle.OracleHydro(frame)
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=9, time_now=0.0, remaining_time=0.2, max_iterations=150)
    assert st.table_fallbacks == rst.table_fallbacks == 0 and st.frame_scatt_cnt == rst.frame_scatt_cnt > 20
    # 1e-6 here, not 1e-9: the reference's formula takes log(1. + 2. * e) of a photon energy e ~ 1e-3 and multiplies it by (1 + e)/e^3 ~ 1e9
    # (mcrat_scattering.c:610-615) -- the rounding of 1 + 2e alone is worth 2e-7 of the result, so two comoving energies that agree to 1e-10 (the
    # engine's and the oracle's do) give cross sections that agree to 1e-7.  Integers (cells, scattering counts, which photon when) are still exact.
    _compare(out, P.aos, rtol=1e-6)
    # ... and it IS the Klein-Nishina factor: tau = n sigma_T sigma_KN(e') (1 - beta cos), so tau / tau_DIRECT = sigma_KN(e')
    d = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    d.set_hydro(frame)
    d.set_photons(ph)
    d.begin_frame(9, 0.0, 0.2)
    d.run(1)
    t = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
    t.set_hot_cross_section(tab)
    t.set_hydro(frame)
    t.set_photons(ph)
    t.begin_frame(9, 0.0, 0.2)
    t.run(1)
    a, b = d.get_photons(), t.get_photons()
    ok = (np.asarray(a["nearest_block_index"]) >= 0) & (np.asarray(a["num_scatt"]) == np.asarray(ph["num_scatt"]))
    L = oracle.lib()
    L.orc_kleinNishinaCrossSection.restype = C.c_double
    kn = np.array([L.orc_kleinNishinaCrossSection(C.c_double(float(x) / (synth.M_EL * synth.C_LIGHT))) for x in np.asarray(a["comv_p0"])[ok]])
    assert ok.sum() > 200 and np.allclose(np.asarray(b["total_optical_depth"])[ok] / np.asarray(a["total_optical_depth"])[ok], kn, rtol=1e-6)
    for x in (e, d, t):
        x.close()


def test_table_mode_needs_its_table(hip):
    frame, ph, cfg = synth.config1(n_photons=100, n0=8, n1=8)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0, tau_calculation=hip.TAU_TABLE)
    e.set_hydro(frame)
    e.set_photons(ph)
    with pytest.raises(hip.McratHipError):
        e.begin_frame(1, 0.0, 0.1)                    # no table yet
    with pytest.raises(hip.McratHipError):
        e.set_hot_cross_section(np.full((5, 5), np.nan))
    d = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    with pytest.raises(hip.McratHipError):
        d.set_hot_cross_section(_hot_table())         # a DIRECT context takes no table


@pytest.mark.parametrize("which", ["flash", "pluto", "sph3d"])
def test_device_built_grid_equals_host_built_grid(hip, which, monkeypatch):
    """the cell-lookup grid is built on the device (grid_build.hip); the host build (MCRAT_HIP_HOST_GRID=1) is its
    cross-check: same answers for interior points, face / corner points and points outside the mesh"""
    rng = np.random.default_rng(11)
    if which == "flash":
        frame, _, cfg = synth.config2(n_photons=64, nzc=16)
    elif which == "pluto":
        frame, _, cfg = synth.config3(n_photons=64, nr=192, nth=96)
    else:
        frame, _, cfg = synth.config_3d(synth.SPHERICAL, n_photons=64)
    three = cfg["dimensions"] == synth.THREE
    M = frame["num_elements"]
    pick = rng.integers(0, M, 20000)
    u = rng.random((3, pick.size)) - 0.5
    snap = (np.arange(pick.size) % 3 == 0)
    u[:, snap] = np.sign(u[:, snap]) * 0.5
    a0 = frame["r0"][pick] + u[0] * frame["r0_size"][pick]
    a1 = frame["r1"][pick] + u[1] * frame["r1_size"][pick]
    a2 = frame["r2"][pick] + u[2] * frame["r2_size"][pick] if three else None
    got = []
    for host_grid in (False, True):
        if host_grid:
            monkeypatch.setenv("MCRAT_HIP_HOST_GRID", "1")
        else:
            monkeypatch.delenv("MCRAT_HIP_HOST_GRID", raising=False)
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
        e.set_hydro(frame)
        got.append(e.lookup_cell(a0, a1, a2))
        e.close()
    assert np.array_equal(got[0], got[1])
    assert (got[0][~snap] == pick[~snap]).all()


# ------------------------------------------------------------------ photonInjection on the device
@pytest.mark.parametrize("case", ["cfg2-bb", "cfg2-wien", "cfg3-spherical", "3d-cartesian"])
def test_photon_injection_equals_oracle(hip, oracle, case):
    """mcrat_hip_inject_photons against orc_photonInjection on the same frame and seed: the same number of photons (the
    Poisson counts and the weight loop of mclib.c:87-136), the same weight, the same photons in the same order"""
    if case.startswith("cfg2"):
        frame, _, cfg = synth.config2(n_photons=64, nzc=16)
        args = dict(r_inj=1e12, theta_min=0.0, theta_max=np.radians(3.0))
    elif case == "cfg3-spherical":
        frame, _, cfg = synth.config3(n_photons=64, nr=256, nth=128)
        args = dict(r_inj=1e12, theta_min=0.0, theta_max=np.radians(6.0))
    else:
        frame, _, cfg = synth.config_3d_cartesian(n_photons=64)
        R = np.sqrt(frame["r0"] ** 2 + frame["r1"] ** 2 + frame["r2"] ** 2)
        args = dict(r_inj=float(np.median(R)), theta_min=0.0, theta_max=np.pi)
    spect = "w" if case == "cfg2-wien" else "b"
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], 0)
    ref, wref = oracle.photon_injection(c, H, args["r_inj"], 1e42, 20000, 60000, spect, args["theta_min"], args["theta_max"], seed=2024)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    e.set_hydro(frame)
    n, w = e.inject_photons(args["r_inj"], 1e42, 20000, 60000, spect, args["theta_min"], args["theta_max"], frame["fps"], 2024)
    assert n == len(ref) and 20000 <= n <= 60000 and w == wref
    out = e.get_photons_aos()
    for k in ("type", "num_scatt", "weight", "nearest_block_index", "recalc_properties", "s0", "s1", "s2", "s3"):
        assert np.array_equal(out[k], ref[k]), k
    for k in ("r0", "r1", "r2", "p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
        scale = np.abs(ref[k])
        if k in ("p1", "p2", "p3"):
            scale = np.abs(ref["p0"])
        if k in ("comv_p1", "comv_p2", "comv_p3"):
            scale = np.abs(ref["comv_p0"])
        if k in ("r0", "r1", "r2"):
            scale = np.maximum(scale, 1e9)
        assert (np.abs(out[k] - ref[k]) <= 1e-11 * scale).all(), k
    # ... and the injected photons run: one forced pass locates every one of them in a cell
    e.begin_frame(5, 0.0, 1.0 / frame["fps"])
    st = e.run(3)
    assert st.iterations == 3 and st.not_found == 0
    assert (e.get_photons()["nearest_block_index"] >= 0).all()


def test_photon_injection_errors(hip):
    frame, _, cfg = synth.config2(n_photons=64, nzc=8)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    with pytest.raises(hip.McratHipError):
        e.inject_photons(1e12, 1e45, 100, 1000, "b", 0.0, 0.05, 5.0, 1)          # no hydro frame staged
    e.set_hydro(frame)
    with pytest.raises(hip.McratHipError):
        e.inject_photons(1e12, 1e45, 100, 1000, "x", 0.0, 0.05, 5.0, 1)          # unknown spectrum
    with pytest.raises(hip.McratHipError):
        e.inject_photons(1e15, 1e45, 100, 1000, "b", 0.0, 0.05, 5.0, 1)          # no cell touches the slab: the count stays 0
