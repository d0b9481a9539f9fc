"""Host-side C mirror (mcrat_amd/host): mc.par grammar and the scatter-frame driver."""
import ctypes as C
import os

import numpy as np
import pytest

from mcrat_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the reference's sample_mc.par content is restated here as the grammar's known-answer case
# (same positional layout as /root/reference/sample_mc.par; values chosen for the test)
SAMPLE = """[Hydro/MHD Simulation Block]

5.               # Number of frames per second of hydro simulation
3000\t\t# Last available hydro simulation frame
0 5e12\t\t# Max r0 coordinate limits of hydro simulation
0 2.5e12\t\t# Max r1 coordinate limit of hydro simulation
0 2e13\t\t# Max r2 coordinate limit of hydro simulation (if simulation is 3D)

[MCRaT Injection Angles Block]

0.               \t# The minimum off-axis angle to inject photons (in degrees)
6.               \t# The maximum off-axis angle to inject photons (in degrees)
3.\t\t\t# Number of angle bins to consider
200 210 220      \t# Frame at which photon injection starts for each angle bin
2 3 4            \t# Number of frames for which photons are injected for each angle bin
1e11 1.5e12 2e12\t# The radius at which the photons are injected for each angle bin

[MCRaT Photon Block]

b\t\t# Type of spectrum we inject with, w=wien b=blackbody
1000\t\t# Min number of photons
5000\t\t# Max number of photons

[Initialization/Continuation Block]

i\t\t# Initialize or continue simulation (i=initialize (delete all files) c=continue)
"""


class McPar(C.Structure):
    _fields_ = [("fps", C.c_double), ("last_frame", C.c_int),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2),
                ("theta_jmin", C.c_double), ("theta_j", C.c_double), ("n_theta_j", C.c_int),
                ("frm0", C.POINTER(C.c_int)), ("frm2", C.POINTER(C.c_int)), ("inj_radius", C.POINTER(C.c_double)),
                ("spect", C.c_char), ("min_photons", C.c_int), ("max_photons", C.c_int), ("restart", C.c_char)]


@pytest.fixture(scope="module")
def host():
    from mcrat_amd import build
    from mcrat_amd.host import build_host
    build.build()
    lib = C.CDLL(build_host.build())
    lib.mcrat_host_read_mcpar.restype = C.c_int
    lib.mcrat_host_read_mcpar.argtypes = [C.c_char_p, C.POINTER(McPar)]
    lib.mcrat_host_free_mcpar.argtypes = [C.POINTER(McPar)]
    return lib


def test_mcpar_grammar(host, tmp_path):
    path = tmp_path / "mc.par"
    path.write_text(SAMPLE)
    p = McPar()
    assert host.mcrat_host_read_mcpar(str(path).encode(), C.byref(p)) == 0
    assert p.fps == 5.0 and p.last_frame == 3000
    assert list(p.r0_domain) == [0.0, 5e12] and list(p.r1_domain) == [0.0, 2.5e12] and list(p.r2_domain) == [0.0, 2e13]
    assert (p.theta_jmin, p.theta_j, p.n_theta_j) == (0.0, 6.0, 3)
    assert [p.frm0[i] for i in range(3)] == [200, 210, 220]
    assert [p.frm2[i] for i in range(3)] == [202, 213, 224]          # start + number of frames, mcrat_io.c:1201
    assert [p.inj_radius[i] for i in range(3)] == [float(np.float32(1e11)), float(np.float32(1.5e12)), float(np.float32(2e12))]
    assert (p.spect, p.min_photons, p.max_photons, p.restart) == (b"b", 1000, 5000, b"i")
    host.mcrat_host_free_mcpar(C.byref(p))
    # errors are codes, not crashes (the reference dereferences a NULL FILE* on a missing file)
    assert host.mcrat_host_read_mcpar(str(tmp_path / "absent.par").encode(), C.byref(p)) == -1
    (tmp_path / "bad.par").write_text(SAMPLE.replace("5000", "").replace("\nb\t", "\nx\t"))
    assert host.mcrat_host_read_mcpar(str(tmp_path / "bad.par").encode(), C.byref(p)) == -2


def test_hot_cross_section_file(host, tmp_path):
    """the file format of createHotCrossSection / readHotCrossSection (hot_x_section.c:107-133, 208-254)"""
    n_e, n_t = 6, 4
    rng = np.random.default_rng(3)
    tab = rng.normal(size=(n_e + 1, n_t + 1))
    lines = ["The comoving photon energy and the temperatures are normalized by the electron rest mass",
             "The calculated hot cross sections are normalized by the thompson cross section.",
             "Photon index\tTheta Index\tlog10(Comoving Photon Energy)\tlog10(Theta)\tlog10(Hot Cross Section)",
             "------------------------------------------------"]
    for i in range(n_e + 1):
        for j in range(n_t + 1):
            lines.append("%d\t%d\t%g\t%g\t%15.10g" % (i, j, -12 + i * 3.0, -4 + j * 2.0, tab[i, j]))
    f = tmp_path / "thermal_hot_x_section.dat"
    f.write_text("\n".join(lines) + "\n")
    host.mcrat_host_read_hot_cross_section.restype = C.c_int
    host.mcrat_host_read_hot_cross_section.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_int]
    out = np.zeros((n_e + 1, n_t + 1))
    assert host.mcrat_host_read_hot_cross_section(str(f).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n_e, n_t) == 0
    assert np.allclose(out, tab, rtol=1e-9, atol=0)               # %15.10g
    # a file larger than the compiled bounds, a truncated file, a missing file
    assert host.mcrat_host_read_hot_cross_section(str(f).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n_e - 1, n_t) == -2
    g = tmp_path / "short.dat"
    g.write_text("\n".join(lines[:-3]) + "\n")
    assert host.mcrat_host_read_hot_cross_section(str(g).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n_e, n_t) == -2
    assert host.mcrat_host_read_hot_cross_section(b"/nonexistent/x.dat", out.ctypes.data_as(C.POINTER(C.c_double)), n_e, n_t) == -1


def test_reference_sample_file_parses_if_present(host):
    ref = "/root/reference/sample_mc.par"          # not present on the GPU box; read as data, not imported
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted here")
    p = McPar()
    assert host.mcrat_host_read_mcpar(ref.encode(), C.byref(p)) == 0
    assert p.fps == 5.0 and p.last_frame == 3000 and p.n_theta_j == 3 and p.min_photons == 1000 and p.max_photons == 5000
    assert [p.frm2[i] - p.frm0[i] for i in range(3)] == [2, 2, 2]
    host.mcrat_host_free_mcpar(C.byref(p))


@pytest.mark.gpu
def test_scatter_frame_driver_matches_engine_and_logs_like_the_reference(host, tmp_path):
    from mcrat_amd import engine
    frame, ph, cfg = synth.config1(n_photons=400, n0=16, n1=16)
    fps, scatt_frame, time_now, seed = 400.0, 7, 7.0 / 400.0, 99        # a 2.5 ms frame: a few hundred events
    frame["fps"] = fps
    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    tn, st = e.propagate_frame(time_now, (scatt_frame + 1) / fps - time_now, seed)
    want = e.get_photons_aos()

    # the same frame through the C driver on caller-owned struct photon records
    e2 = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
    plist = engine.PhotonList(aos.ctypes.data, None, len(aos), 0, len(aos))
    keep, h = [], engine.Hydro()
    h.num_elements = frame["num_elements"]
    for f in ("r0", "r1", "r0_size", "r1_size", "v0", "v1", "dens_lab", "temp", "gamma"):
        a = np.ascontiguousarray(frame[f], dtype=np.float64)
        keep.append(a)
        setattr(h, f, a.ctypes.data_as(C.POINTER(C.c_double)))
    for k in ("r0_domain", "r1_domain"):
        getattr(h, k)[0], getattr(h, k)[1] = frame[k]
    h.fps = fps
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    log = tmp_path / "mc_output_0.log"
    fp = libc.fopen(str(log).encode(), b"a")
    host.mcrat_host_scatter_frame.restype = C.c_int
    host.mcrat_host_scatter_frame.argtypes = [C.c_void_p, C.POINTER(engine.PhotonList), C.POINTER(engine.Hydro), C.POINTER(C.c_double),
                                              C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.POINTER(engine.FrameStats)]
    t = C.c_double(time_now)
    st2 = engine.FrameStats()
    rc = host.mcrat_host_scatter_frame(e2.ctx, C.byref(plist), C.byref(h), C.byref(t), scatt_frame, 1, fps, seed, fp, C.byref(st2))
    libc.fclose(fp)
    assert rc == 0
    assert t.value == tn == pytest.approx((scatt_frame + 1) / fps, rel=1e-13)
    assert st2.frame_scatt_cnt == st.frame_scatt_cnt > 50 and st2.iterations == st.iterations
    for k in aos.dtype.names:
        assert np.array_equal(aos[k], want[k]), k
    text = log.read_text()
    assert "The number of scatterings in this frame is: %d\n" % st.frame_scatt_cnt in text
    assert "MCRaT had to refind the position of photons %d times in this frame.\n" % st.num_photons_find_new_element in text
    assert "The maximum number of scatterings for a photon is: %d\n" % int(want["num_scatt"].max()) in text
    assert "The average number of scatterings thus far is: %f\n" % want["num_scatt"].mean() in text


@pytest.mark.gpu
def test_ab_shims_run_mains_loop_function_by_function(host, oracle):
    """the loop of mcrat.c:761-851 written as main() writes it, once on the reference-signature shims
    (mcrat_ab_findContainingHydroCell / calcMeanFreePath / photonEvent / updatePhotonPosition) and once on the oracle's
    restatements of the same functions: the same decisions and photons after every call"""
    from mcrat_amd import engine
    frame, ph, cfg = synth.config2(n_photons=600, nzc=8, lumi=1e54)
    seed, time_now0, remaining0 = 4242, 1.0, 2.0e-3               # a 2 ms stretch of the frame: a few dozen passes, then the frame ends

    class Ab(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("last_rc", C.c_int), ("relocated_seen", C.c_longlong), ("scatt_seen", C.c_longlong)]

    PL, HY = C.POINTER(engine.PhotonList), C.POINTER(engine.Hydro)
    host.mcrat_ab_begin_frame.restype = C.c_int
    host.mcrat_ab_begin_frame.argtypes = [C.POINTER(Ab), C.c_void_p, PL, HY, C.c_uint64, C.c_double, C.c_double]
    host.mcrat_ab_findContainingHydroCell.restype = C.c_int
    host.mcrat_ab_findContainingHydroCell.argtypes = [PL, HY, C.c_int, C.POINTER(Ab), C.c_void_p]
    host.mcrat_ab_calcMeanFreePath.restype = None
    host.mcrat_ab_calcMeanFreePath.argtypes = [PL, HY, C.POINTER(Ab), C.c_void_p]
    host.mcrat_ab_photonEvent.restype = C.c_double
    host.mcrat_ab_photonEvent.argtypes = [PL, C.c_double, HY, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Ab), C.c_void_p]
    host.mcrat_ab_updatePhotonPosition.restype = None
    host.mcrat_ab_updatePhotonPosition.argtypes = [PL, C.c_double, C.POINTER(Ab), C.c_void_p]

    e = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
    sorted_idx = np.zeros(len(aos), dtype=np.int32)
    plist = engine.PhotonList(aos.ctypes.data, sorted_idx.ctypes.data_as(C.POINTER(C.c_int)), len(aos), 0, len(aos))
    keep, h = [], engine.Hydro()
    h.num_elements = frame["num_elements"]
    for f in ("r0", "r1", "r0_size", "r1_size", "v0", "v1", "dens_lab", "temp", "gamma"):
        a = np.ascontiguousarray(frame[f], dtype=np.float64)
        keep.append(a)
        setattr(h, f, a.ctypes.data_as(C.POINTER(C.c_double)))
    for k in ("r0_domain", "r1_domain"):
        getattr(h, k)[0], getattr(h, k)[1] = frame[k]
    h.fps = frame["fps"]
    ab = Ab()
    assert host.mcrat_ab_begin_frame(C.byref(ab), e.ctx, C.byref(plist), C.byref(h), seed, time_now0, remaining0) == 0

    # the oracle side, function by function (orc_* restate the same reference functions)
    L = oracle.lib()
    oc = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    H = oracle.OracleHydro(frame)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    rng = oracle.Rng()
    L.orc_rng_init(C.byref(rng), seed, 0)
    ost = oracle.Stats()

    def same_photons(tag):
        for k in ("nearest_block_index", "num_scatt", "recalc_properties"):
            assert np.array_equal(aos[k], P.aos[k]), (tag, k)
        for k in ("r0", "r1", "r2", "p0", "comv_p0", "time_to_scatter", "total_optical_depth"):
            assert np.allclose(aos[k], P.aos[k], rtol=1e-9, atol=0), (tag, k)

    # mcrat.c:754-851
    find_switch, remaining, o_remaining = 1, remaining0, remaining0
    time_now = o_time_now = time_now0
    scatt_cnt, o_scatt_cnt = C.c_int(0), C.c_longlong(0)
    idx, o_idx = C.c_int(-1), C.c_int(-1)
    relocated = o_relocated = passes = 0
    while remaining > 0:
        L.orc_rng_set_iteration(C.byref(rng), passes)
        n = host.mcrat_ab_findContainingHydroCell(C.byref(plist), C.byref(h), find_switch, C.byref(ab), None)
        assert ab.last_rc == 0
        o_n = L.orc_findContainingHydroCell(C.byref(oc), C.byref(P.c), C.byref(H.c), find_switch, C.byref(ost))
        assert n == o_n
        relocated += n
        host.mcrat_ab_calcMeanFreePath(C.byref(plist), C.byref(h), C.byref(ab), None)
        L.orc_calcMeanFreePath(C.byref(oc), C.byref(P.c), C.byref(H.c), C.byref(rng))
        o_sorted = np.ctypeslib.as_array(P.c.sorted_indexes, shape=(len(aos),))
        assert np.array_equal(sorted_idx[:8], o_sorted[:8])
        same_photons("after the free-path draw of pass %d" % passes)
        find_switch = 0
        if aos["time_to_scatter"][sorted_idx[0]] < remaining:                       # mcrat.c:777
            dt = host.mcrat_ab_photonEvent(C.byref(plist), remaining, C.byref(h), C.byref(idx), C.byref(scatt_cnt), None, C.byref(ab), None)
            assert ab.last_rc == 0
            o_dt = L.orc_photonEvent(C.byref(oc), C.byref(P.c), o_remaining, C.byref(H.c), C.byref(o_idx), C.byref(o_scatt_cnt), C.byref(rng), C.byref(ost))
            assert dt == pytest.approx(o_dt, rel=1e-12) and idx.value == o_idx.value and scatt_cnt.value == o_scatt_cnt.value
            time_now += dt; remaining -= dt                                          # mcrat.c:782-784
            o_time_now += o_dt; o_remaining -= o_dt
        else:                                                                        # mcrat.c:834-845
            host.mcrat_ab_updatePhotonPosition(C.byref(plist), remaining, C.byref(ab), None)
            assert ab.last_rc == 0
            L.orc_updatePhotonPosition(C.byref(P.c), o_remaining)
            time_now += remaining; remaining = 0
            o_time_now += o_remaining; o_remaining = 0
        same_photons("after the event of pass %d" % passes)
        passes += 1
        assert passes < 400
    assert passes > 10 and scatt_cnt.value > 10 and relocated > 0
    assert time_now == pytest.approx(time_now0 + remaining0, rel=1e-13)


def test_hot_cross_section_file_written_like_the_reference_reads_back(host, tmp_path):
    """mcrat_host_write_hot_cross_section writes createHotCrossSection's file format (hot_x_section.c:109-132); the reader of
    readHotCrossSection's format gets the table back to the ten digits the format keeps"""
    host.mcrat_host_write_hot_cross_section.restype = C.c_int
    host.mcrat_host_write_hot_cross_section.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
    host.mcrat_host_read_hot_cross_section.restype = C.c_int
    host.mcrat_host_read_hot_cross_section.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    table = -rng.random((221, 81)) * 3
    path = str(tmp_path / "thermal_hot_x_section.dat").encode()
    assert host.mcrat_host_write_hot_cross_section(path, table.ctypes.data_as(C.POINTER(C.c_double)), 220, 80, -12.0, 6.0, -4.0, 4.0) == 0
    lines = open(path).read().split("\n")
    assert lines[3].startswith("----") and lines[4].split("\t")[:2] == ["0", "0"] and len(lines) == 4 + 221 * 81 + 1
    assert lines[4 + 81].split("\t")[:4] == ["1", "0", "%g" % (-12.0 + 18.0 / 220), "-4"]
    back = np.empty_like(table)
    assert host.mcrat_host_read_hot_cross_section(path, back.ctypes.data_as(C.POINTER(C.c_double)), 220, 80) == 0
    assert np.allclose(back, table, rtol=1e-9, atol=1e-12)
    assert host.mcrat_host_write_hot_cross_section(str(tmp_path / "no" / "x.dat").encode(), table.ctypes.data_as(C.POINTER(C.c_double)), 220, 80, -12.0, 6.0, -4.0, 4.0) == -1
