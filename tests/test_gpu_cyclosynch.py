"""GPU parity for the cyclo-synchrotron row as far as it is on the device: the absorption at the end of a
scatter frame (mcrat_hip_absorb_cyclosynch = phAbsCyclosynch, Src/mc_cyclosynch.c:1571-1623) against oracle/oracle_cyclosynch.c."""
import ctypes as C

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu
PL_CONST = 6.6260755e-27


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


@pytest.mark.parametrize("b_field_calc", [0, 1, 2])
@pytest.mark.parametrize("dims", ["2d", "3d"])
def test_absorption_matches_the_oracle(hip, oracle, b_field_calc, dims):
    L = oracle.lib()
    if dims == "2d":
        frame, ph, cfg = synth.config2(n_photons=4000, nzc=8, stokes=1, lumi=1e53)
    else:
        frame, ph, cfg = synth.config_3d_cartesian(n_photons=3000, n=(16, 16, 16))
    rng = np.random.default_rng(8)
    m = frame["num_elements"]
    dens = np.ascontiguousarray(frame["dens"]) if "dens" in frame else np.ascontiguousarray(frame["dens_lab"] / frame["gamma"])
    B = [np.ascontiguousarray(10 ** rng.uniform(2, 7, m)) for _ in range(3)]
    # locate the photons on the device, then give the list every type and a spread of comoving energies around nu_c
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(3, 0.0, 0.2)
    e.run(40)
    aos = e.get_photons_aos()
    n = len(aos)
    aos["type"] = rng.choice([b"i", b"k", b"c", b"p", b"N"], size=n, p=[0.4, 0.2, 0.2, 0.15, 0.05])
    aos["weight"] = np.where(aos["type"] == b"N", 0.0, 10 ** rng.uniform(45, 47, n))
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    H = oracle.OracleHydro(frame)
    cs = oracle.CS(b_field_calc, 0.5, 0.1, dens.ctypes.data_as(C.POINTER(C.c_double)), *[b.ctypes.data_as(C.POINTER(C.c_double)) for b in B], 200, 200, 0.5, 10.0)
    cell = np.maximum(aos["nearest_block_index"], 0)
    nu_c = np.array([L.orc_calcCyclotronFreq(L.orc_getMagneticFieldMagnitude(C.byref(c), C.byref(cs), C.byref(H.c), int(k))) for k in cell])
    factor = np.where(rng.random(n) < 0.5, 10 ** rng.uniform(-2, -0.01, n), 10 ** rng.uniform(0.01, 2, n))
    factor[::50] = 1.0                                                  # exactly at the cyclotron frequency: absorbed (<=)
    aos["comv_p0"] = factor * nu_c * PL_CONST / synth.C_LIGHT
    aos["nearest_block_index"][5::97] = -1                              # outside the frame: never absorbed
    # oracle
    l = oracle.PhotonList()
    L.orc_list_init(C.byref(l))
    ref = aos.copy()
    assert L.orc_list_set(C.byref(l), ref.ctypes.data, n) == 0
    n_abs, n_scatt = C.c_int(), C.c_int()
    w_ref = L.orc_phAbsCyclosynch(C.byref(c), C.byref(cs), C.byref(l), C.byref(H.c), C.byref(n_abs), C.byref(n_scatt))
    buf = (C.c_char * (n * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    want = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    L.orc_list_free(C.byref(l))
    # device
    e.set_photons_aos(aos)
    e.set_hydro_extras(dens, *B)
    a, s, w = e.absorb_cyclosynch(b_field_calc, 0.5)
    got = e.get_photons_aos()
    e.close()
    assert (a, s) == (n_abs.value, n_scatt.value) and a > 0.3 * n and s > 0
    assert w == pytest.approx(w_ref, rel=1e-12)
    assert (got["type"][got["nearest_block_index"] != -1] == b"p").sum() == 0          # pool photons outside the frame are left alone (:1586)
    for f in ("type", "weight", "nearest_block_index", "recalc_properties", "p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3",
              "r0", "r1", "r2", "s0", "s1", "s2", "s3", "num_scatt", "total_optical_depth"):
        assert np.array_equal(got[f], want[f], equal_nan=got[f].dtype.kind == "f"), f


def _oracle_pool(oracle, frame, cfg, aos, null_slots, dens, b_field_calc, seed, maximum_photons, frames=(200, 200), B=(None, None, None), ph_weight=1e40):
    """-> (photons emitted, weight, fallback flag, the list before the emission, the list after it)"""
    L = oracle.lib()
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    H = oracle.OracleHydro(frame)
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    cs = oracle.CS(b_field_calc, 0.5, 0.1, ptr(dens), ptr(B[0]), ptr(B[1]), ptr(B[2]), frames[0], frames[1], 0.5, 10.0)
    l = oracle.PhotonList()
    L.orc_list_init(C.byref(l))
    a = aos.copy()
    assert L.orc_list_set(C.byref(l), a.ctypes.data, len(a)) == 0
    for i in null_slots:
        assert L.orc_list_set_null(C.byref(l), int(i)) == 0
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    before = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    rng = oracle.Rng()
    L.orc_rng_init(C.byref(rng), seed, 0)
    w, fb = C.c_double(), C.c_int()
    n = L.orc_photonEmitCyclosynch(C.byref(c), C.byref(cs), C.byref(l), 1e12, ph_weight, maximum_photons, 0.0, 0.05, C.byref(H.c), C.byref(rng), 0, 0,
                                   C.byref(w), C.byref(fb))
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    out = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    L.orc_list_free(C.byref(l))
    return n, w.value, fb.value, before, out


@pytest.mark.parametrize("field", [1e12, 1e14])
def test_pool_emission_where_qags_goes_past_its_first_rule(hip, oracle, field):
    """gsl_integration_qags of the Planck photon density up to the cyclotron frequency (mc_cyclosynch.c:1276) in cells whose cyclotron frequency
    lies far beyond the Planck peak (B_FIELD_CALC == SIMULATION with a magnetar's field: h nu_c / kT of 10^2 and more): the 21-point rule on the
    whole interval fails QAGS' first-step test, and the device bisects the worst interval exactly as the oracle's orc_qags does (its documented
    stand-in for the rest of QAGS) -- the same Poisson means, so the same pool photons at the same weight."""
    frame, ph, cfg = synth.config2(n_photons=600, nzc=8, stokes=1, lumi=1e53)
    dens = np.ascontiguousarray(frame["dens"])
    M = frame["num_elements"]
    B = [np.full(M, field), np.zeros(M), np.zeros(M)]
    aos = synth.photons_to_aos(ph, hip.PHOTON_DTYPE)
    # (the weight the search starts from: near the answer -- from far below it the reference's int photon total overflows on the way, :1244-1296)
    n_ref, w_ref, fb_ref, aos, want = _oracle_pool(oracle, frame, cfg, aos, range(1, 600, 2), dens, 2, 77, 1500, B=B, ph_weight=1e50)
    assert fb_ref == 1 and n_ref > 0                        # the oracle went past the first rule
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_hydro_extras(dens, *B)
    e.set_photons_aos(aos)
    n, w, bad = e.emit_cyclosynch_pool(1e12, 1e50, 1500, 0.0, 0.05, frame["fps"], 77, b_field_calc=2, scatt_frame_number=200, inj_frame_number=200)
    assert (n, w, bad) == (n_ref, w_ref, 0)
    got = e.get_photons_aos()
    assert np.array_equal(got["type"], want["type"]) and np.array_equal(got["weight"], want["weight"])
    pool = got["type"] == b"p"
    assert pool.sum() == n
    for f in ("p0", "r0", "r1", "r2"):
        scale = np.abs(want["p0"][pool]) if f == "p0" else 1e12
        assert np.all(np.abs(got[f][pool] - want[f][pool]) <= 1e-11 * scale), f
    # (a field so strong that every one of the rule's 21 nodes lies beyond the Planck peak's reach integrates to exactly 0 with error 0 -- QAGS, the
    # oracle and the device all call that converged; the device's 64-interval limit is a safety net this integrand does not reach)
    e.close()


@pytest.mark.parametrize("case", ["null-slots", "full-list-doubles", "no-cell-in-shell"])
def test_pool_emission_matches_the_oracle(hip, oracle, case):
    """mcrat_hip_emit_cyclosynch_pool against orc_photonEmitCyclosynch (inject_single_switch = 0): the same number of pool photons, the
    same adjusted weight, in the same slots; directions and positions to 1e-11 (the device boosts with the staged cell records)"""
    frame, ph, cfg = synth.config2(n_photons=600, nzc=8, stokes=1, lumi=1e53)
    dens = np.ascontiguousarray(frame["dens"])
    aos = synth.photons_to_aos(ph, hip.PHOTON_DTYPE)
    nulls = range(1, 600, 2) if case != "full-list-doubles" else ()
    frames = (100000, 200) if case == "no-cell-in-shell" else (200, 200)
    maximum = 1500 if case != "full-list-doubles" else 9000       # no null slot: the list doubles (photons.c:112-121)
    n_ref, w_ref, fb_ref, aos, want = _oracle_pool(oracle, frame, cfg, aos, nulls, dens, 1, 77, maximum, frames)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_hydro_extras(dens)
    e.set_photons_aos(aos)
    n, w, bad = e.emit_cyclosynch_pool(1e12, 1e40, maximum, 0.0, 0.05, frame["fps"], 77, b_field_calc=1, scatt_frame_number=frames[0], inj_frame_number=frames[1])
    assert (n, w, bad) == (n_ref, w_ref, 0) and fb_ref == 0
    assert e.n == len(want)
    got = e.get_photons_aos()
    e.close()
    if case == "no-cell-in-shell":
        assert n == 0
    elif case == "full-list-doubles":
        assert len(got) == 1200 and n > 0
    else:
        assert 1 <= n <= 150
    assert np.array_equal(got["type"], want["type"]) and np.array_equal(got["weight"], want["weight"])
    assert np.array_equal(got["nearest_block_index"], want["nearest_block_index"]) and np.array_equal(got["recalc_properties"], want["recalc_properties"])
    pool = got["type"] == b"p"
    assert pool.sum() == n
    for f in ("comv_p0",):
        assert np.allclose(got[f][pool], want[f][pool], rtol=1e-14)
    for f in ("p0", "p1", "p2", "p3", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2"):
        scale = np.abs(want["p0"][pool]) if f.startswith("p") else (np.abs(want["comv_p0"][pool]) if f.startswith("comv") else 1e12)
        assert np.all(np.abs(got[f][pool] - want[f][pool]) <= 1e-11 * scale), f
    rest = ~pool
    for f in ("p0", "r0", "num_scatt", "s0"):
        assert np.array_equal(got[f][rest], want[f][rest], equal_nan=True), f


def _rebin_input(dtype, n, seed, null_every=0):
    rng = np.random.default_rng(seed)
    aos = np.zeros(n, dtype=dtype)
    kind = rng.choice([b"k", b"c", b"i", b"p"], size=n, p=[0.5, 0.3, 0.15, 0.05])
    aos["type"] = kind
    th_pos, phi_pos, r = rng.uniform(0.01, 0.04, n), rng.uniform(0, 2 * np.pi, n), rng.uniform(1.0e12, 1.1e12, n)
    aos["r0"], aos["r1"], aos["r2"] = r * np.sin(th_pos) * np.cos(phi_pos), r * np.sin(th_pos) * np.sin(phi_pos), r * np.cos(th_pos)
    e = 10 ** rng.uniform(-18, -15, n)
    th_d, ph_d = rng.uniform(0.0, 0.05, n), phi_pos + rng.normal(0, 0.01, n)
    aos["p0"], aos["p1"], aos["p2"], aos["p3"] = e, e * np.sin(th_d) * np.cos(ph_d), e * np.sin(th_d) * np.sin(ph_d), e * np.cos(th_d)
    aos["s0"], aos["s1"], aos["s2"] = 1.0, rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)
    aos["weight"] = 10 ** rng.uniform(45, 47, n)
    aos["num_scatt"] = rng.integers(1, 40, n)
    aos["nearest_block_index"] = 5
    return aos


@pytest.mark.parametrize("dims,max_photons,e_perc,ang_phi", [("2d", 2000, 0.1, 10.0), ("3d", 5000, 0.01, 45.0)])
def test_rebinning_matches_the_oracle(hip, oracle, dims, max_photons, e_perc, ang_phi):
    """mcrat_hip_rebin_cyclosynch against orc_rebinCyclosynchCompPhotons (mc_cyclosynch.c:246-712): the same bins, the same empty-bin
    count and counters, every rebinned photon in the same slot; the sums are formed in slot order on both sides so weights agree to the
    last bit and the trigonometric fields to 1e-13"""
    L = oracle.lib()
    D = synth.TWO if dims == "2d" else synth.THREE
    n = 3000
    aos = _rebin_input(oracle.PHOTON_DTYPE, n, 4)
    c = oracle.make_config(D, synth.CYLINDRICAL if dims == "2d" else synth.CARTESIAN, 1)
    cs = oracle.CS(1, 0.5, e_perc, None, None, None, None, 200, 200, 0.5, ang_phi)
    l = oracle.PhotonList()
    L.orc_list_init(C.byref(l))
    a = aos.copy()
    assert L.orc_list_set(C.byref(l), a.ctypes.data, len(a)) == 0
    for i in range(0, n, 7):                                    # some null slots to start with
        assert L.orc_list_set_null(C.byref(l), int(i)) == 0
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    before = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    emit, scatt = C.c_int(), C.c_int()
    nulls = L.orc_rebinCyclosynchCompPhotons(C.byref(c), C.byref(cs), C.byref(l), C.byref(emit), C.byref(scatt), max_photons)
    assert nulls >= 0
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    want = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    L.orc_list_free(C.byref(l))

    e = hip.Engine(D, synth.CYLINDRICAL if dims == "2d" else synth.CARTESIAN, 1)
    e.set_photons_aos(before.astype(hip.PHOTON_DTYPE))
    got_nulls, got_emit, got_scatt = e.rebin_cyclosynch(max_photons, e_perc, 0.5, ang_phi)
    got = e.get_photons_aos()
    e.close()
    assert (got_nulls, got_emit, got_scatt) == (nulls, emit.value, scatt.value)
    assert scatt.value > 100
    assert np.array_equal(got["type"], want["type"])
    for f in ("weight", "num_scatt", "nearest_block_index", "recalc_properties", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
        assert np.array_equal(got[f], want[f]), f
    k = want["type"] == b"k"
    for f in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3"):
        scale = np.abs(want["p0"]) if f.startswith("p") else (1e12 if f.startswith("r") else 1.0)
        assert np.all(np.abs(got[f] - want[f]) <= 1e-13 * scale), (f, np.max(np.abs(got[f] - want[f]) / scale))
        assert np.array_equal(got[f][~k], want[f][~k]), f


def test_rebinning_refusals(hip, oracle):
    """the reference's error paths: more bins than max_photons (:649-654), nothing to rebin (:640-645)"""
    aos = _rebin_input(hip.PHOTON_DTYPE, 500, 9)
    e = hip.Engine(synth.TWO, synth.CYLINDRICAL, 1)
    e.set_photons_aos(aos)
    with pytest.raises(hip.McratHipError):
        e.rebin_cyclosynch(10, 0.5, 0.01, 10.0)                 # 5 energy bins x hundreds of angle bins > 10
    only = aos.copy()
    only["type"] = b"i"
    e.set_photons_aos(only)
    with pytest.raises(hip.McratHipError):
        e.rebin_cyclosynch(2000)
    assert np.array_equal(e.get_photons_aos()["p0"], only["p0"])
    e.close()


CS_FRAMES = {
    # name: (mesh, B_FIELD_CALC, max_photons, remaining_time, max_iterations, emit_pool, theta_max, rebin_ang_phi)
    "2d-pool-replace-grow": ("2d", 1, 2000, 0.2, 600, 1, 0.05, 10.0),
    "2d-rebin-in-the-loop": ("2d", 1, 200, 3.0, 0, 1, 0.05, 10.0),
    "2d-internal-energy-field": ("2d", 0, 2000, 0.2, 800, 1, 0.05, 10.0),
    "2d-first-frame-no-pool": ("2d", 1, 2000, 0.2, 800, 0, 0.05, 10.0),
    "3d-pool-replace-grow": ("3d", 1, 2000, 0.2, 0, 1, 0.06, 45.0),
    "3d-internal-energy-field": ("3d", 0, 2000, 0.2, 0, 1, 0.06, 45.0),
    "3d-rebin-refused": ("3d", 1, 150, 1.0, 0, 1, 0.06, 45.0),
    # BASELINE.json configs[4]'s switches at test size: THREE / SPHERICAL, B_FIELD_CALC == SIMULATION, STOKES on; the list doubles twice
    "3d-spherical-simulation-field": ("3ds", 2, 2000, 0.2, 0, 1, 0.2, 45.0),
}


@pytest.mark.parametrize("case", sorted(CS_FRAMES))
def test_scatter_frame_with_the_switch_on_matches_the_oracle(hip, oracle, case):
    """mcrat_hip_scatter_frame_cyclosynch against orc_scatter_frame_cs (mcrat.c:706-878): pool emission, the loop in which every scattered
    pool photon becomes a comptonised one and is replaced (the list doubling when its null slots run out), the rebinning every 1000
    scatterings once there are more comptonised photons than max_photons (refused, as in the reference, when it would need more bins
    than max_photons), the absorption at the end.  Same passes, scatterings, counters, list length, types, slots and weights; doubles
    to 1e-9 as for the plain loop (tests/test_gpu_parity.py)."""
    mesh, b_field_calc, max_photons, remaining, max_iterations, emit_pool, theta_max, ang_phi = CS_FRAMES[case]
    L = oracle.lib()
    if mesh == "2d":
        frame, ph, cfg = synth.config2(n_photons=300, nzc=8, lumi=3e53)
    elif mesh == "3d":
        frame, ph, cfg = synth.config_3d_cartesian(n_photons=300, n=(8, 8, 8))
    else:
        frame, ph, cfg = synth.config_3d(synth.SPHERICAL, n_photons=300)
    dens = np.ascontiguousarray(frame["dens"])
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], 1)
    H = oracle.OracleHydro(frame)
    B = [None, None, None]
    if b_field_calc == 2:
        g = np.random.default_rng(5)
        B = [np.ascontiguousarray(g.uniform(1e3, 1e5, frame["num_elements"])) for _ in range(3)]
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    cs = oracle.CS(b_field_calc, 0.5, 0.1, ptr(dens), ptr(B[0]), ptr(B[1]), ptr(B[2]), 200, 200, 0.5, ang_phi)
    aos = synth.photons_to_aos(ph, oracle.PHOTON_DTYPE)
    l = oracle.PhotonList()
    L.orc_list_init(C.byref(l))
    both = np.concatenate([aos, aos])                          # the second half becomes the null slots the pool goes into
    assert L.orc_list_set(C.byref(l), both.ctypes.data, len(both)) == 0
    for i in range(300, 600):
        assert L.orc_list_set_null(C.byref(l), i) == 0
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    before = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    rng = oracle.Rng()
    L.orc_rng_init(C.byref(rng), 31, 0)
    st, cnt, t = oracle.Stats(), oracle.CSCounts(), C.c_double(0.0)
    L.orc_scatter_frame_cs(C.byref(c), C.byref(cs), C.byref(l), C.byref(H.c), C.byref(rng), C.byref(t), remaining, 1e12, 1e40, max_photons, 0.0, theta_max,
                           emit_pool, max_iterations, C.byref(st), C.byref(cnt))
    assert cnt.error == 0 and st.frame_scatt_cnt > 500
    buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
    want = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
    L.orc_list_free(C.byref(l))
    if case == "3d-spherical-simulation-field":
        assert len(want) == 2400
    if case.endswith("pool-replace-grow"):
        assert len(want) == 1200 and cnt.scatt_cyclosynch_num_ph > 0 and cnt.frame_abs_cnt > 0   # the list doubled inside the loop
    if case == "2d-rebin-in-the-loop":
        assert cnt.rebins >= 1 and st.remaining_time == 0.0 and len(want) == 1200
    if case == "3d-rebin-refused":
        assert cnt.rebins == 0 and cnt.num_cyclosynch_ph_emit > max_photons
    if not emit_pool:
        assert cnt.num_cyclosynch_ph_emit == 0 and (want["type"] == b"p").sum() == 0

    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 1, cyclosynchrotron=1)
    e.set_hydro(frame)
    e.set_hydro_extras(dens, *B)
    e.set_photons_aos(before.astype(hip.PHOTON_DTYPE))
    with pytest.raises(hip.McratHipError):                     # the plain loop has no hook: refused with the switch on
        e.propagate_frame(0.0, remaining, 31)
    tn, gst, gcnt = e.scatter_frame_cyclosynch(0.0, remaining, 31, 1e12, 1e40, max_photons, 0.0, theta_max, frame["fps"], emit_pool=emit_pool,
                                               max_iterations=max_iterations, b_field_calc=b_field_calc, rebin_ang_phi=ang_phi,
                                               scatt_frame_number=200, inj_frame_number=200)
    got = e.get_photons_aos()
    e.close()
    assert (gst.iterations, gst.frame_scatt_cnt, gst.kn_rejections) == (st.iterations, st.frame_scatt_cnt, st.kn_rejections)
    assert (gcnt.num_cyclosynch_ph_emit, gcnt.scatt_cyclosynch_num_ph, gcnt.frame_abs_cnt, gcnt.rebins) == \
        (cnt.num_cyclosynch_ph_emit, cnt.scatt_cyclosynch_num_ph, cnt.frame_abs_cnt, cnt.rebins)
    assert gcnt.pool_weight == cnt.pool_weight and gcnt.n_comptonized == pytest.approx(cnt.n_comptonized, rel=1e-12)
    assert tn == pytest.approx(t.value, rel=1e-12) and len(got) == len(want)
    assert np.array_equal(got["type"], want["type"])
    for f in ("weight", "num_scatt", "nearest_block_index", "recalc_properties"):
        assert np.array_equal(got[f], want[f]), f
    for f in ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3"):
        scale = np.maximum(np.abs(want["p0"]), 1e-300) if f.startswith("p") else (
            np.maximum(np.abs(want["comv_p0"]), 1e-300) if f.startswith("comv") else (np.maximum(np.abs(want[f]), 1e9) if f.startswith("r") else 1.0))
        err = np.abs(got[f] - want[f]) / scale
        assert np.all(err <= 1e-9), (f, float(err.max()), int(err.argmax()))
