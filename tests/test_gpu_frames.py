"""The scatter-frame loop of main() across its three device-side stages (Src/mcrat.c:633-905), end to end on the GPU against
the oracle doing the same steps: injection frame ingested (ph_inj_switch = 1) -> photonInjection -> for each hydro frame:
phMinMax -> getHydroData with the photons' slab (ph_inj_switch = 0) -> the photon loop for 1/fps -> frame statistics; then
the output columns and the checkpoint.  Every stage is compared where the reference hands data from one to the next."""
import ctypes as C

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu

DOM = dict(r0_domain=(1e11, 4e12), r1_domain=(0.0, 0.6), r2_domain=(0.0, 0.0))


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _close(a, b, rtol, scale=None):
    a, b = np.asarray(a, float), np.asarray(b, float)
    s = np.maximum(np.abs(b), 1e-300) if scale is None else scale
    bad = np.abs(a - b) > rtol * s
    assert not bad.any(), (int(bad.sum()), a[bad][:3], b[bad][:3])


@pytest.mark.parametrize("kind", ["pluto", "chombo"])
def test_three_hydro_frames_like_main(hip, oracle, kind):
    fps, r_inj, frames = 5.0, 1e12, 3
    if kind == "pluto":
        raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    else:
        raw = synth.chombo_raw(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (128, 32), seed=31, logr=True, refine_below=(0.6, 0.4))
    jet = dict(lumi=2e53, theta_j=0.1)                       # dense enough for hundreds of scatterings per frame
    cfg = oracle.make_config(synth.TWO, synth.SPHERICAL, 1)
    e = hip.Engine(synth.TWO, synth.SPHERICAL, 1)

    # ---- the injection frame (mcrat.c:633-645)
    inj = dict(r_inj=r_inj, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=fps)
    n_cells, ef, _ = e.ingest(raw, dict(inj, **DOM), hip.Engine.outflow(3, **jet))
    ref_frame, ef_ref = oracle.hydro_ingest(cfg, raw, inj, oracle.outflow(3, **jet))
    assert (n_cells, ef) == (ref_frame["num_elements"], ef_ref)
    n, w = e.inject_photons(r_inj, 1e50, 800, 1600, "b", 0.0, 0.08, fps, seed=404)
    H = oracle.OracleHydro(dict(ref_frame, **DOM, fps=fps))
    ref_ph, w_ref = oracle.photon_injection(cfg, H, r_inj, 1e50, 800, 1600, "b", 0.0, 0.08, seed=404)
    assert (n, w) == (len(ref_ph), w_ref)
    injected = e.get_photons_aos()
    _close(injected["r0"], ref_ph["r0"], 1e-11)
    _close(injected["p0"], ref_ph["p0"], 1e-11)
    # from here on both sides carry the same list: a 1e-11 difference at injection would be amplified by the boosts
    P = oracle.OraclePhotons(injected)

    time_now, total_scatt = 0.0, 0
    for k in range(frames):
        # ---- phMinMax -> getHydroData(ph_inj_switch = 0) (mcrat.c:704-721)
        mm = e.ph_minmax()
        ref_mm = [C.c_double() for _ in range(4)]
        oracle.lib().orc_phMinMax(C.byref(P.c), *[C.byref(x) for x in ref_mm])
        _close(mm[:2], [x.value for x in ref_mm[:2]], 1e-13)
        # theta = acos(z / r) near the axis is ill-conditioned (an ulp of z/r moves a 1e-3 angle by 1e-10 of itself)
        assert abs(mm[2] - ref_mm[2].value) < 1e-12 and abs(mm[3] - ref_mm[3].value) < 1e-12
        slab = dict(r_inj=r_inj, ph_inj_switch=0, min_r=mm[0], max_r=mm[1], min_theta=mm[2], max_theta=mm[3], fps=fps)
        n_cells, ef, _ = e.ingest(raw, dict(slab, **DOM), hip.Engine.outflow(3, **jet))
        ref_frame, ef_ref = oracle.hydro_ingest(cfg, raw, slab, oracle.outflow(3, **jet))
        assert (n_cells, ef) == (ref_frame["num_elements"], ef_ref) and n_cells < (raw["nx"] * raw["ny"] if kind == "pluto" else 10 ** 9)
        # ---- the loop for this hydro frame (mcrat.c:754-851)
        remaining = (k + 1) / fps - time_now
        new_time, st = e.propagate_frame(time_now, remaining, seed=1000 + k)
        H = oracle.OracleHydro(dict(ref_frame, **DOM, fps=fps))
        ost, otime, orem, _ = oracle.photon_loop(cfg, P, H, seed=1000 + k, time_now=time_now, remaining_time=remaining)
        assert (st.iterations, st.frame_scatt_cnt, st.num_photons_find_new_element) == (ost.iterations, ost.frame_scatt_cnt, ost.num_photons_find_new_element)
        assert new_time == pytest.approx(otime, rel=1e-14) and orem <= 0
        time_now = new_time
        total_scatt += st.frame_scatt_cnt
        # ---- phScattStats (mcrat.c:881)
        mx, mn, avg, ravg = e.scatt_stats()
        r = [C.c_int(), C.c_int(), C.c_double(), C.c_double()]
        oracle.lib().orc_phScattStats(C.byref(P.c), *[C.byref(x) for x in r])
        assert (mx, mn) == (r[0].value, r[1].value) and avg == pytest.approx(r[2].value, rel=1e-12) and ravg == pytest.approx(r[3].value, rel=1e-9)
    assert total_scatt > 300

    # ---- the lists agree after three frames: integers exact, doubles to 1e-9
    out = e.get_photons_aos()
    assert np.array_equal(out["nearest_block_index"], P.aos["nearest_block_index"])
    assert np.array_equal(out["num_scatt"], P.aos["num_scatt"])
    p0 = np.abs(P.aos["p0"])
    for k in ("p0", "p1", "p2", "p3"):
        _close(out[k], P.aos[k], 1e-9, p0)
    for k in ("r0", "r1", "r2"):
        _close(out[k], P.aos[k], 1e-9, np.maximum(np.abs(P.aos[k]), 1e9))
    for k in ("s0", "s1", "s2", "s3"):
        _close(out[k], P.aos[k], 1e-9, np.ones(len(out)))
    # ---- what printPhotons would write
    cols = e.get_output()
    keep = P.aos["weight"] != 0
    assert len(cols["p0"]) == int(keep.sum())
    _close(cols["comv_p0"], P.aos["comv_p0"][keep], 1e-9, np.abs(P.aos["comv_p0"][keep]))
    e.close()


def test_resident_frame_driver_of_the_host_c_equals_the_steps_done_by_hand(hip, tmp_path):
    """mcrat_host_scatter_frame_resident (host C): phMinMax -> the caller's reader (a callback that ends in
    mcrat_hip_ingest_pluto) -> the loop -> statistics and the reference's log lines, photons resident throughout"""
    from mcrat_amd.host import build_host
    host = C.CDLL(build_host.build())
    GET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(hip.Slab))
    host.mcrat_host_scatter_frame_resident.restype = C.c_int
    host.mcrat_host_scatter_frame_resident.argtypes = [C.c_void_p, GET, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double, C.c_uint64,
                                                       C.c_void_p, C.POINTER(hip.FrameStats)]
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    fps, r_inj = 5.0, 1e12
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (384, 96), seed=31, log_axis0=True)
    jet = hip.Engine.outflow(3, lumi=2e53, theta_j=0.1)
    inj = dict(r_inj=r_inj, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=fps, **DOM)

    def start():
        e = hip.Engine(synth.TWO, synth.SPHERICAL, 1)
        e.ingest(raw, inj, jet)
        e.inject_photons(r_inj, 1e50, 800, 1600, "b", 0.0, 0.08, fps, seed=404)
        return e

    # by hand
    a = start()
    t_a, stats_a = 0.0, []
    for k in range(3):
        mm = a.ph_minmax()
        a.ingest(raw, dict(r_inj=r_inj, ph_inj_switch=0, min_r=mm[0], max_r=mm[1], min_theta=mm[2], max_theta=mm[3], fps=fps, **DOM), jet)
        t_a, st = a.propagate_frame(t_a, (k + 1) / fps - t_a, 1000 + k)
        stats_a.append((st.iterations, st.frame_scatt_cnt, a.num_elements))
    out_a = a.get_photons_aos()
    a.close()

    # through the host C driver
    b = start()
    seen = []

    def reader(user, ctx, scatt_frame, slab):
        s = slab.contents
        seen.append((scatt_frame, s.ph_inj_switch, s.min_r, s.max_r))
        n, ef, _ = b.ingest(raw, dict(r_inj=s.r_inj, ph_inj_switch=s.ph_inj_switch, min_r=s.min_r, max_r=s.max_r, min_theta=s.min_theta,
                                      max_theta=s.max_theta, fps=s.fps, r0_domain=tuple(s.r0_domain), r1_domain=tuple(s.r1_domain),
                                      r2_domain=tuple(s.r2_domain)), jet)
        return 0
    cb = GET(reader)
    log = libc.fopen(str(tmp_path / "mc_output_0.log").encode(), b"w")
    t_b = C.c_double(0.0)
    d0, d1, d2 = (C.c_double * 2)(*DOM["r0_domain"]), (C.c_double * 2)(*DOM["r1_domain"]), (C.c_double * 2)(*DOM["r2_domain"])
    stats_b = []
    for k in range(3):
        st = hip.FrameStats()
        assert host.mcrat_host_scatter_frame_resident(b.ctx, cb, None, r_inj, d0, d1, d2, C.byref(t_b), k, 1, fps, 1000 + k, log, C.byref(st)) == 0
        stats_b.append((st.iterations, st.frame_scatt_cnt, b.num_elements))
    libc.fclose(log)
    out_b = b.get_photons_aos()
    b.close()
    assert stats_a == stats_b and t_b.value == t_a and [s[:2] for s in seen] == [(0, 0), (1, 0), (2, 0)]
    for k in out_a.dtype.names:
        assert np.array_equal(out_a[k], out_b[k], equal_nan=out_a[k].dtype.kind == "f"), k
    text = (tmp_path / "mc_output_0.log").read_text()
    assert text.count("The number of scatterings in this frame is: ") == 3
    assert "The number of scatterings in this frame is: %d\n" % stats_b[0][1] in text
    assert "MCRaT had to refind the position of photons" in text and "The average position of photons is" in text
    # a failing reader stops the frame with its code
    bad = GET(lambda user, ctx, frame, slab: -7)
    c = start()
    assert host.mcrat_host_scatter_frame_resident(c.ctx, bad, None, r_inj, d0, d1, d2, C.byref(t_b), 3, 1, fps, 1, None, None) == -7
    c.close()


def test_forty_frames_of_the_structured_jet_stay_consistent(hip):
    """a short version of tools/lundman_run.py (the manual's global validation set-up): photons injected below the photosphere
    of a structured jet, forty hydro frames of phMinMax -> slab ingest -> loop as virtual ranks, Stokes on.  Size-independent
    checks: the clock lands on every frame boundary, the photons' scattering counts add up to the frames' event counters,
    everything stays finite and physical, the selected slab follows the photons outwards."""
    fps, r_inj = 5.0, 3e10
    dom = dict(r0_domain=(1e9, 2.5e13), r1_domain=(0.0, np.pi / 2), r2_domain=(0.0, 0.0))
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e9, 0.0), (2.5e13, np.pi / 2), (1024, 256), seed=1, log_axis0=True)
    jet = hip.Engine.outflow(hip.STRUCTURED_SPHERICAL_OUTFLOW, lumi=3e50, theta_j=0.1, p=4.0)
    e = hip.Engine(synth.TWO, synth.SPHERICAL, 1, virtual_rank_photons=500)
    e.ingest(raw, dict(r_inj=r_inj, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=fps, **dom), jet)
    n, w = e.inject_photons(r_inj, 1e48, 10000, 20000, "b", 0.0, 0.08, fps, seed=2014)
    t, events, r_mean, cells = 0.0, 0, [], []
    for k in range(40):
        mm = e.ph_minmax()
        m, ef, _ = e.ingest(raw, dict(r_inj=r_inj, ph_inj_switch=0, min_r=mm[0], max_r=mm[1], min_theta=mm[2], max_theta=mm[3], fps=fps, **dom), jet)
        t, st = e.propagate_frame(t, (k + 1) / fps - t, 700 + k)
        assert t == pytest.approx((k + 1) / fps, rel=1e-14) and st.remaining_time == 0.0
        events += st.frame_scatt_cnt
        r_mean.append(e.scatt_stats()[3])
        cells.append(m)
        assert ef == 1 and m > 0
    out = e.get_photons_aos()
    assert int(out["num_scatt"].sum()) == events and events > 5 * n           # tau ~ 6 at injection on the axis
    assert (np.diff(r_mean) > 0).all() and r_mean[-1] == pytest.approx(r_inj + 39.5 / fps * synth.C_LIGHT, rel=0.05)   # streaming outwards at ~c
    for k in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3"):
        assert np.isfinite(out[k]).all(), k
    assert np.allclose(np.sqrt(out["p1"] ** 2 + out["p2"] ** 2 + out["p3"] ** 2), out["p0"], rtol=1e-12)     # null 4-momenta
    assert (out["s0"] == 1).all() and (np.hypot(out["s1"], out["s2"]) <= 1 + 1e-9).all() and (out["s3"] == 0).all()
    assert (out["weight"] == w).all() and (out["type"] == b"i").all()
    assert cells[-1] < cells[0]                                                 # a narrower shell of the log-r mesh as the pulse thins out
    e.close()
