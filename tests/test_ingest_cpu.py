"""CPU checks for the hydro-ingest row: known answers that pin the oracle's restatement of the readers
(oracle/oracle_ingest.c) against independent numpy constructions, and the host-side PLUTO file parser
(mcrat_amd/host, mcrat_host_read_pluto).  The reference ships no fixtures for its readers."""
import ctypes as C
import os

import numpy as np
import pytest

from mcrat_amd import synth


def _slab(**kw):
    d = dict(r_inj=1e12, ph_inj_switch=0, min_r=0.995e12, max_r=1.002e12, min_theta=0.0, max_theta=0.04, fps=5.0)
    d.update(kw)
    return d


def test_flash_expansion_is_the_block_mesh_and_parents_are_skipped(oracle):
    side = 2.5e8 * 8
    z_lo = 1e12 - 8 * side
    raw = synth.flash_raw_blocks(side, 8, 16, 8, z_lo, seed=4)
    assert (raw["node_type"] == 2).sum() > 10
    cfg = oracle.make_config(synth.TWO, synth.CYLINDRICAL, 0)
    cols, ef = oracle.hydro_ingest(cfg, raw, _slab(ph_inj_switch=1, r_inj=0.0))       # r > 0: every leaf cell
    mesh = synth.flash_like_mesh(side, 8, 16, 8, z_lo, (0, 5e12), (0, 2.5e13), 5.0)  # built cell by cell, no blocks
    assert ef == 1 and cols["num_elements"] == mesh["num_elements"] == 64 * int((raw["node_type"] == 1).sum())
    for k in ("r0", "r1", "r0_size", "r1_size"):
        assert np.allclose(cols[k], mesh[k], rtol=1e-15, atol=0), k
    leaf = raw["node_type"] == 1
    assert np.array_equal(cols["v0"], raw["velx"][leaf].ravel()) and np.array_equal(cols["v1"], raw["vely"][leaf].ravel())
    assert np.array_equal(cols["dens"], raw["dens"][leaf].ravel() * raw["d_scale"])
    assert np.array_equal(cols["pres"], raw["pres"][leaf].ravel() * raw["p_scale"])
    v2 = cols["v0"] ** 2 + cols["v1"] ** 2
    assert np.allclose(cols["gamma"], 1 / np.sqrt(1 - v2), rtol=1e-15) and np.allclose(cols["dens_lab"], cols["dens"] * cols["gamma"], rtol=1e-14)
    assert np.allclose(cols["temp"], (3 * cols["pres"] / synth.A_RAD) ** 0.25, rtol=1e-14)
    assert np.allclose(cols["r"], np.hypot(cols["r0"], cols["r1"]), rtol=1e-15) and np.allclose(cols["theta"], np.arctan2(cols["r0"], cols["r1"]), rtol=1e-14)


def _corners_2d(geometry, x0, x1, s0, s1):
    lo0, lo1, hi0, hi1 = x0 - 0.5 * s0, x1 - 0.5 * s1, x0 + 0.5 * s0, x1 + 0.5 * s1
    if geometry == synth.SPHERICAL:
        return lo0, lo1, hi0, hi1
    return np.hypot(lo0, lo1), np.arctan2(lo0, lo1), np.hypot(hi0, hi1), np.arctan2(hi0, hi1)


@pytest.mark.parametrize("kind", ["flash", "pluto-spherical"])
def test_slab_selection_is_the_corner_test_in_cell_order(oracle, kind):
    """mclib_flash.c:288-318 / mclib_pluto.c:1260-1301: inner-corner / outer-corner radii and angles against the photons'
    slab widened by elem_factor light-frames and 2 degrees; kept cells stay in reader order"""
    if kind == "flash":
        geom = synth.CYLINDRICAL
        raw = synth.flash_raw_blocks(2e9, 8, 16, 8, 1e12 - 1.6e10, seed=1, parent_every=0)
        cfg = oracle.make_config(synth.TWO, geom, 0)
        everything, _ = oracle.hydro_ingest(cfg, raw, _slab(ph_inj_switch=1, r_inj=0.0))
    else:
        geom = synth.SPHERICAL
        raw = synth.pluto_raw_grid(synth.TWO, geom, (1e11, 0.0), (4e12, 0.6), (96, 40), seed=1, log_axis0=True)
        cfg = oracle.make_config(synth.TWO, geom, 0)
        everything, _ = oracle.hydro_ingest(cfg, raw, _slab(ph_inj_switch=1, r_inj=0.0))
        assert everything["num_elements"] == 96 * 40
        X2, X1 = np.meshgrid(raw["x2"], raw["x1"] * raw["l_scale"], indexing="ij")
        assert np.array_equal(everything["r0"], X1.ravel()) and np.array_equal(everything["r1"], X2.ravel())     # x1 fastest
    s = _slab()
    got, ef = oracle.hydro_ingest(cfg, raw, s)
    r_in, th_in, r_out, th_out = _corners_2d(geom, everything["r0"], everything["r1"], everything["r0_size"], everything["r1_size"])
    margin = ef * synth.C_LIGHT / s["fps"]
    deg2 = 2 * 0.017453292519943295
    keep = (s["min_r"] - margin <= r_out) & (r_in <= s["max_r"] + margin) & (th_out >= s["min_theta"] - deg2) & (th_in <= s["max_theta"] + deg2)
    assert ef == 1 and 0 < keep.sum() < keep.size
    assert got["num_elements"] == int(keep.sum())
    for k in ("r0", "r1", "r0_size", "r1_size", "v0", "v1", "dens", "pres"):
        assert np.array_equal(got[k], everything[k][keep]), k
    # injection frames: centre radius beyond 0.95 r_inj
    got, _ = oracle.hydro_ingest(cfg, raw, _slab(ph_inj_switch=1, r_inj=1.03e12))
    keep = everything["r"] > 0.95 * 1.03e12
    assert got["num_elements"] == int(keep.sum()) and np.array_equal(got["r0"], everything["r0"][keep])


def test_elem_factor_widens_the_slab_one_light_frame_at_a_time(oracle):
    raw = synth.pluto_raw_grid(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (1e12, 0.6), (64, 16), seed=1, log_axis0=True)
    cfg = oracle.make_config(synth.TWO, synth.SPHERICAL, 0)
    step = synth.C_LIGHT / 5.0
    for k in (1, 2, 5):
        s = _slab(min_r=1e12 + (k - 0.5) * step, max_r=1e12 + (k - 0.5) * step + 1e8)
        cols, ef = oracle.hydro_ingest(cfg, raw, s)
        assert ef == k and cols["num_elements"] > 0
        assert (cols["r0"] + 0.5 * cols["r0_size"]).max() <= 1e12 * (1 + 1e-12)
    with pytest.raises(RuntimeError):
        oracle.hydro_ingest(cfg, raw, _slab(min_theta=2.0, max_theta=2.1), max_elem_factor=20)


@pytest.mark.parametrize("geom", [synth.CYLINDRICAL, synth.SPHERICAL])
def test_analytic_outflows_follow_the_closed_forms(oracle, geom):
    """analytic_outflows.c against the numpy formulas of mcrat_amd.synth (written from the manual's equations)"""
    if geom == synth.CYLINDRICAL:
        raw = synth.flash_raw_blocks(2e9, 8, 16, 8, 1e12 - 1.6e10, seed=1)
    else:
        raw = synth.pluto_raw_grid(synth.TWO, geom, (1e9, 0.0), (4e12, 0.6), (96, 40), seed=1, log_axis0=True)
    cfg = oracle.make_config(synth.TWO, geom, 0)
    s = _slab(ph_inj_switch=1, r_inj=0.0)
    base, _ = oracle.hydro_ingest(cfg, raw, s)
    frame = dict(base, dimensions=synth.TWO, geometry=geom)

    sph, _ = oracle.hydro_ingest(cfg, raw, s, oracle.outflow(2))
    ref = synth.spherical_outflow(dict(frame), gamma_infinity=100.0, lumi=1e54, r00=1e8)
    for k in ("gamma", "pres", "dens", "dens_lab", "temp", "v0", "v1"):
        assert np.allclose(sph[k], ref[k], rtol=1e-12, atol=1e-300), k
    coast = sph["r"] >= 1e10
    assert np.allclose(sph["temp"][coast] * sph["r"][coast] ** (2.0 / 3.0), (sph["temp"][coast] * sph["r"][coast] ** (2.0 / 3.0))[0], rtol=1e-12)

    jet, _ = oracle.hydro_ingest(cfg, raw, s, oracle.outflow(3, lumi=3e50, theta_j=0.1))
    ref = synth.structured_fireball(dict(frame), gamma_0=100.0, lumi=3e50, r00=1e8, theta_j=0.1, p=4.0)
    ok = jet["gamma"] >= 1                                   # inside r_sat the reference's gamma = r / r_sat < 1 gives NaN velocities
    assert ok.sum() > 0.9 * ok.size
    for k in ("gamma", "pres", "dens", "dens_lab", "temp", "v0", "v1"):
        assert np.allclose(jet[k][ok], ref[k][ok], rtol=1e-12, atol=1e-300), k
    assert jet["gamma"].max() <= 100 and (jet["gamma"][jet["theta"] >= 0.1 * 50 ** 0.25] <= 2.0).all()

    cyl, _ = oracle.hydro_ingest(cfg, raw, s, oracle.outflow(1))
    vel = np.sqrt(1 - 1e-4)
    assert (cyl["gamma"] == 100).all() and (cyl["temp"] == 1e5).all() and (cyl["dens"] == 3e-7).all() and np.allclose(cyl["dens_lab"], 3e-5)
    assert np.allclose(cyl["pres"], synth.A_RAD * 1e20 / 3, rtol=1e-14)
    if geom == synth.CYLINDRICAL:
        assert (cyl["v0"] == 0).all() and np.allclose(cyl["v1"], vel)
    else:                                                    # along the axis, in the (r, theta) basis
        assert np.allclose(cyl["v0"], vel * np.cos(cyl["r1"])) and np.allclose(cyl["v1"], -vel * np.sin(cyl["r1"]))


@pytest.mark.parametrize("case", ["2d-cylindrical", "3d-spherical-logr"])
def test_chombo_levels_boxes_and_the_covered_cell_mask(oracle, case):
    """readPlutoChombo (mclib_pluto.c:12-801): an injection frame (mask on) tiles the domain exactly once with cells of all
    levels; coordinates follow domBeg + dx (i + 1/2) or the logarithmic radial rule; the photons'-slab branch keeps covered
    coarse cells (reference behaviour)"""
    if case == "2d-cylindrical":
        dims, geom, lo, hi, n0, logr = synth.TWO, synth.CYLINDRICAL, (0.0, 8e11), (4e11, 1.6e12), (32, 64), False
    else:
        dims, geom, lo, hi, n0, logr = synth.THREE, synth.SPHERICAL, (2e11, 0.0, 0.0), (3e12, 0.8, 2 * np.pi), (32, 16, 16), True
    raw = synth.chombo_raw(dims, geom, lo, hi, n0, seed=2, logr=logr)
    cfg = oracle.make_config(dims, geom, 0)
    cells = sum(len(lv["data"]) for lv in raw["levels"]) // len(raw["var_names"])
    tiled, ef = oracle.hydro_ingest(cfg, raw, _slab(ph_inj_switch=1, r_inj=0.0))
    assert ef == 1 and tiled["num_elements"] < cells
    lo0, hi0 = tiled["r0"] - 0.5 * tiled["r0_size"], tiled["r0"] + 0.5 * tiled["r0_size"]
    if logr:                                            # cell edges are lo * exp(dx i): measure in ln r
        m0, total0 = np.log(hi0 / lo0), np.log(hi[0] / lo[0])
        assert np.allclose(tiled["r0"], 0.5 * (lo0 + hi0), rtol=1e-14)
    else:
        m0, total0 = tiled["r0_size"], hi[0] - lo[0]
    measure = m0 * tiled["r1_size"] * (tiled["r2_size"] if dims == synth.THREE else 1.0)
    total = total0 * (hi[1] - lo[1]) * ((hi[2] - lo[2]) if dims == synth.THREE else 1.0)
    assert np.isclose(measure.sum(), total, rtol=1e-11)
    assert len(np.unique(np.round(tiled["r1_size"] / tiled["r1_size"].min()))) == 3      # three levels, ratio 2
    assert lo0.min() >= lo[0] * (1 - 1e-12) and hi0.max() <= hi[0] * (1 + 1e-12)
    # every cell of the file in file order: level 0 first, boxes of 8^d cells, x fastest
    allc, _ = oracle.hydro_ingest(cfg, raw, _slab(min_r=0.0, max_r=1e14, min_theta=0.0, max_theta=3.2))
    assert allc["num_elements"] == cells
    lv0 = raw["levels"][0]
    n_lv0 = len(lv0["data"]) // len(raw["var_names"])
    assert len(np.unique(allc["r1_size"][:n_lv0])) == 1 and allc["r1_size"][:n_lv0][0] == allc["r1_size"].max()
    assert (np.diff(allc["r0"][:8]) > 0).all() and allc["r1"][0] == allc["r1"][7]
    k = raw["var_names"].index("rho")
    nd = 3 if dims == synth.THREE else 2
    assert np.array_equal(allc["dens"][:8 ** nd], lv0["data"][k * 8 ** nd:(k + 1) * 8 ** nd] * raw["d_scale"])     # variable-major per box


# ---------------------------------------------------------------------------------------------- PLUTO files
class PlutoGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int)] + \
               [(f, C.POINTER(C.c_double)) for f in ("x1", "dx1", "x2", "dx2", "x3", "dx3", "rho", "vx1", "vx2", "vx3", "prs")] + \
               [("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double)]


class HostPluto(C.Structure):
    _fields_ = [("grid", PlutoGrid), ("num_vars", C.c_int), ("var_names", C.POINTER(C.c_char_p)),
                ("axes", C.POINTER(C.c_double)), ("data", C.POINTER(C.c_double))]


def write_pluto_files(directory, raw, frame, var_order, three):
    """grid.out, dbl.out and data.NNNN.dbl as PLUTO 4 writes them (single_file, little endian)"""
    n = [raw["nx"], raw["ny"]] + ([raw["nz"]] if three else [])
    axes = [(raw["x1"], raw["dx1"]), (raw["x2"], raw["dx2"])] + ([(raw["x3"], raw["dx3"])] if three else [])
    lines = ["# ******************************************************", "# PLUTO 4.3 Grid File", "# Generated on  Sat Oct  3 2026",
             "#", "# DIMENSIONS: %d" % len(n), "# GEOMETRY:   SPHERICAL"]
    edges = []
    for a, (c, w) in enumerate(axes):
        left, right = c - 0.5 * w, c + 0.5 * w
        edges.append((left, right))
        lines.append("# X%d: [ %.6e,  %.6e], %d point(s), 0 ghosts" % (a + 1, left[0], right[-1], n[a]))
    lines.append("# ******************************************************")
    for a, (left, right) in enumerate(edges):
        lines.append("%d" % n[a])
        lines += [" %d   %.12e    %.12e" % (i + 1, left[i], right[i]) for i in range(n[a])]
    if not three:
        lines += ["1", " 1   0.000000000000e+00    1.000000000000e+00"]
    with open(os.path.join(directory, "grid.out"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(directory, "dbl.out"), "w") as f:
        f.write("%d %.6e %.6e %d single_file little %s \n" % (frame, 0.2 * frame, 1e-4, 100 * frame, " ".join(var_order)))
        f.write("%d %.6e %.6e %d single_file little %s \n" % (frame + 1, 0.2 * (frame + 1), 1e-4, 100 * (frame + 1), " ".join(var_order)))
    blocks = [np.ascontiguousarray(raw[v], dtype="<f8").ravel() for v in var_order]
    np.concatenate(blocks).tofile(os.path.join(directory, "data.%04d.dbl" % frame))
    return edges


@pytest.fixture(scope="module")
def host():
    from mcrat_amd import build
    from mcrat_amd.host import build_host
    build.build()
    lib = C.CDLL(build_host.build())
    lib.mcrat_host_read_pluto.restype = C.c_int
    lib.mcrat_host_read_pluto.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(HostPluto)]
    lib.mcrat_host_free_pluto.argtypes = [C.POINTER(HostPluto)]
    lib.mcrat_host_pluto_name.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
    return lib


@pytest.mark.parametrize("three", [False, True])
def test_pluto_files_parse_into_the_readers_buffers(host, oracle, tmp_path, three):
    if three:
        dims, order = synth.THREE, ["rho", "vx1", "vx2", "vx3", "prs", "tr1"]
        raw = synth.pluto_raw_grid(dims, synth.SPHERICAL, (2e11, 0.0, 0.0), (3e12, 0.5, 2 * np.pi), (24, 10, 6), seed=3, log_axis0=True)
        raw["tr1"] = np.full_like(raw["rho"], 7.0)             # a variable the reader does not use
    else:
        dims, order = synth.TWO, ["rho", "prs", "vx1", "vx2"]  # picked by name, not by position
        raw = synth.pluto_raw_grid(dims, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.6), (48, 20), seed=3, log_axis0=True)
    edges = write_pluto_files(str(tmp_path), raw, 37, order, three)
    name = C.create_string_buffer(512)
    host.mcrat_host_pluto_name(name, 512, os.path.join(str(tmp_path), "data.").encode(), 37)
    assert name.value.decode().endswith("data.0037.dbl")
    for frame, expect in ((3, "0003"), (12, "0012"), (345, "0345"), (2024, "2024")):    # mclib_pluto.c:829-844
        host.mcrat_host_pluto_name(name, 512, b"x.", frame)
        assert name.value == ("x.%s.dbl" % expect).encode()
    host.mcrat_host_pluto_name(name, 512, os.path.join(str(tmp_path), "data.").encode(), 37)

    hp = HostPluto()
    rc = host.mcrat_host_read_pluto(os.path.join(str(tmp_path), "grid.out").encode(), os.path.join(str(tmp_path), "dbl.out").encode(),
                                    name.value, int(three), raw["l_scale"], raw["d_scale"], raw["p_scale"], C.byref(hp))
    assert rc == 0
    g = hp.grid
    assert (g.nx, g.ny, g.nz) == (raw["nx"], raw["ny"], raw["nz"] if three else 1)
    assert [hp.var_names[i].decode() for i in range(hp.num_vars)] == order
    ax = ("x1", "x2", "x3")[:3 if three else 2]
    parsed = dict(kind="pluto", nx=g.nx, ny=g.ny, nz=g.nz, l_scale=g.l_scale, d_scale=g.d_scale, p_scale=g.p_scale)
    for a, k in enumerate(ax):
        n = (g.nx, g.ny, g.nz)[a]
        left = np.array([float("%.12e" % v) for v in edges[a][0]])      # what the text file holds
        right = np.array([float("%.12e" % v) for v in edges[a][1]])
        c = np.ctypeslib.as_array(getattr(g, k), shape=(n,)).copy()
        w = np.ctypeslib.as_array(getattr(g, "d" + k), shape=(n,)).copy()
        assert np.array_equal(c, 0.5 * (left + right)) and np.array_equal(w, right - left), k     # mclib_pluto.c:951-971
        parsed[k], parsed["d" + k] = c, w
    cells = g.nx * g.ny * g.nz
    for v in ("rho", "vx1", "vx2", "prs") + (("vx3",) if three else ()):
        parsed[v] = np.ctypeslib.as_array(getattr(g, v), shape=(cells,)).copy()
        assert np.array_equal(parsed[v], np.asarray(raw[v]).ravel()), v
    if not three:
        assert not g.vx3 and not g.x3
    # the parsed buffers select like the originals (up to the 12 digits of the text grid)
    cfg = oracle.make_config(dims, synth.SPHERICAL, 0)
    a, ef_a = oracle.hydro_ingest(cfg, parsed, _slab(max_theta=0.2))
    b, ef_b = oracle.hydro_ingest(cfg, raw, _slab(max_theta=0.2))
    assert ef_a == ef_b and abs(a["num_elements"] - b["num_elements"]) <= 2 and a["num_elements"] > 10
    host.mcrat_host_free_pluto(C.byref(hp))

    # malformed inputs are reported, never read past
    short = tmp_path / "short.dbl"
    short.write_bytes(b"\0" * 64)
    assert host.mcrat_host_read_pluto(os.path.join(str(tmp_path), "grid.out").encode(), os.path.join(str(tmp_path), "dbl.out").encode(),
                                      str(short).encode(), int(three), 1.0, 1.0, 1.0, C.byref(hp)) == -2
    assert host.mcrat_host_read_pluto(b"/nonexistent/grid.out", b"x", b"y", 0, 1.0, 1.0, 1.0, C.byref(hp)) == -1
    novar = tmp_path / "novar.out"
    novar.write_text("0 0.0 1e-4 0 single_file little rho vx1 vx2\n")
    assert host.mcrat_host_read_pluto(os.path.join(str(tmp_path), "grid.out").encode(), str(novar).encode(), name.value, int(three),
                                      1.0, 1.0, 1.0, C.byref(hp)) == -2
