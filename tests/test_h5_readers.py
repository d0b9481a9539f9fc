"""The HDF5 file reads of the host C (mcrat_amd/host/mcrat_hip_host_h5.c: mcrat_host_read_flash, mcrat_host_read_chombo) on real
files with the layout readAndDecimate / readPlutoChombo open; the files are written by a test-only C helper
(tests/data/h5_fixture_writer.c) from mcrat_amd.synth's buffers.  CPU-only: what the readers return is compared with what
went in, and fed to the oracle's selection."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcrat_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def h5(tmp_path_factory):
    from mcrat_amd import build, engine
    from mcrat_amd.host import build_host
    build.build()
    path = build_host.build_h5()
    if path is None:
        pytest.skip("no HDF5 C library in this image")
    prefix = build_host.hdf5_prefix()
    out = str(tmp_path_factory.mktemp("h5w") / "libh5fixture.so")
    libdir = next(os.path.join(prefix, d) for d in ("lib", "lib64", "lib/x86_64-linux-gnu") if os.path.exists(os.path.join(prefix, d, "libhdf5.so")))
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(prefix, "include"),
                    os.path.join(ROOT, "tests", "data", "h5_fixture_writer.c"), "-o", out, "-L", libdir, "-lhdf5", "-Wl,-rpath," + libdir], check=True)
    lib = C.CDLL(path)
    lib.writer = C.CDLL(out)
    lib.engine = engine

    class HostFlash(C.Structure):
        _fields_ = [("blocks", engine.FlashBlocks)]

    class HostChombo(C.Structure):
        _fields_ = [("frame", engine.Chombo), ("levels", C.POINTER(engine.ChomboLevel)), ("var_names", C.POINTER(C.c_char_p)), ("data", C.POINTER(C.c_double))]
    lib.HostFlash, lib.HostChombo = HostFlash, HostChombo
    lib.mcrat_host_read_flash.restype = C.c_int
    lib.mcrat_host_read_flash.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_double, C.POINTER(HostFlash)]
    lib.mcrat_host_read_chombo.restype = C.c_int
    lib.mcrat_host_read_chombo.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(HostChombo)]
    lib.mcrat_host_flash_name.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
    return lib


def test_flash_checkpoint_file(h5, oracle, tmp_path):
    e = h5.engine
    raw = synth.flash_raw_blocks(2e9, 8, 16, 8, 1e12 - 1.6e10, seed=12)
    keep = []

    def ptr(a, dtype=np.float64, ctype=C.c_double):
        arr = np.ascontiguousarray(a, dtype=dtype)
        keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(ctype))
    n = len(raw["node_type"])
    b = e.FlashBlocks(n, 3, 2, ptr(raw["coordinates"]), ptr(raw["block_size"]), ptr(raw["node_type"], np.int32, C.c_int), ptr(raw["velx"]),
                      ptr(raw["vely"]), ptr(raw["dens"]), ptr(raw["pres"]), 1.0, 1.0, 1.0)
    name = C.create_string_buffer(600)
    h5.mcrat_host_flash_name(name, 600, (str(tmp_path) + "/grb_hdf5_plt_cnt_").encode(), 42)
    assert name.value.decode().endswith("grb_hdf5_plt_cnt_0042")                       # modifyFlashName, mclib_flash.c:15-58
    assert h5.writer.fixture_write_flash(name.value, C.byref(b)) == 0
    hf = h5.HostFlash()
    assert h5.mcrat_host_read_flash(name.value, raw["l_scale"], raw["d_scale"], raw["p_scale"], C.byref(hf)) == 0
    g = hf.blocks
    assert (g.n_blocks, g.coord_stride, g.bsize_stride, g.l_scale, g.p_scale) == (n, 3, 2, raw["l_scale"], raw["p_scale"])
    back = dict(kind="flash", l_scale=g.l_scale, d_scale=g.d_scale, p_scale=g.p_scale,
                coordinates=np.ctypeslib.as_array(g.coordinates, shape=(n, 3)).copy(), block_size=np.ctypeslib.as_array(g.block_size, shape=(n, 2)).copy(),
                node_type=np.ctypeslib.as_array(g.node_type, shape=(n,)).copy())
    for k in ("velx", "vely", "dens", "pres"):
        back[k] = np.ctypeslib.as_array(getattr(g, k), shape=(n, 1, 8, 8)).copy()
        assert np.array_equal(back[k], raw[k]), k
    assert np.array_equal(back["coordinates"], raw["coordinates"]) and np.array_equal(back["node_type"], raw["node_type"])
    cfg = oracle.make_config(synth.TWO, synth.CYLINDRICAL, 0)
    slab = dict(r_inj=1e12, ph_inj_switch=0, min_r=0.995e12, max_r=1.002e12, min_theta=0.0, max_theta=0.04, fps=5.0)
    a, _ = oracle.hydro_ingest(cfg, back, slab)
    c, _ = oracle.hydro_ingest(cfg, raw, slab)
    assert a["num_elements"] == c["num_elements"] > 100 and np.array_equal(a["r0"], c["r0"]) and np.array_equal(a["temp"], c["temp"])
    h5.mcrat_host_free_flash(C.byref(hf))
    assert h5.mcrat_host_read_flash(b"/nonexistent", 1.0, 1.0, 1.0, C.byref(hf)) == -1
    # a file without the variables is refused
    b2 = e.FlashBlocks(n, 3, 2, ptr(raw["coordinates"]), ptr(raw["block_size"]), ptr(raw["node_type"], np.int32, C.c_int), ptr(raw["velx"]),
                       ptr(raw["vely"]), ptr(raw["dens"]), ptr(raw["pres"]), 1.0, 1.0, 1.0)
    other = str(tmp_path / "chombo_not_flash.h5").encode()
    raw_c = synth.chombo_raw(synth.TWO, synth.SPHERICAL, (1e11, 0.0), (4e12, 0.8), (32, 16), seed=1, logr=True)
    from oracle import oracle_py
    hc = oracle_py.fill_chombo(raw_c, e.ChomboLevel, e.Chombo, keep)
    assert h5.writer.fixture_write_chombo(other, 0, C.byref(hc)) == 0
    assert h5.mcrat_host_read_flash(other, 1.0, 1.0, 1.0, C.byref(hf)) == -2


@pytest.mark.parametrize("three", [False, True])
def test_pluto_chombo_file(h5, oracle, tmp_path, three):
    from oracle import oracle_py
    e = h5.engine
    if three:
        dims, geom = synth.THREE, synth.SPHERICAL
        raw = synth.chombo_raw(dims, geom, (2e11, 0.0, 0.0), (3e12, 0.8, 2 * np.pi), (32, 16, 16), seed=5, logr=True)
    else:
        dims, geom = synth.TWO, synth.CYLINDRICAL
        raw = synth.chombo_raw(dims, geom, (0.0, 8e11), (4e11, 1.6e12), (32, 64), seed=5)
    keep = []
    h = oracle_py.fill_chombo(raw, e.ChomboLevel, e.Chombo, keep)          # same member layout as mcrat_hip_chombo
    path = str(tmp_path / "data.0007.hdf5").encode()
    assert h5.writer.fixture_write_chombo(path, int(three), C.byref(h)) == 0
    hc = h5.HostChombo()
    assert h5.mcrat_host_read_chombo(path, int(three), raw["l_scale"], raw["d_scale"], raw["p_scale"], C.byref(hc)) == 0
    f = hc.frame
    assert (f.num_levels, f.num_vars) == (len(raw["levels"]), len(raw["var_names"]))
    assert [f.var_names[k].decode() for k in range(f.num_vars)] == raw["var_names"]
    back = dict(kind="chombo", var_names=raw["var_names"], l_scale=f.l_scale, d_scale=f.d_scale, p_scale=f.p_scale, levels=[])
    off = 0
    bi = 6 if three else 4
    for i, lv in enumerate(raw["levels"]):
        L = f.levels[i]
        assert (L.n_boxes, L.data_len, L.ref_ratio, L.logr) == (len(lv["boxes"]), len(lv["data"]), 2, lv["logr"])
        assert (L.dx, L.dombeg1, L.dombeg2, L.g_x2stretch) == (lv["dx"], lv["dombeg1"], lv["dombeg2"], lv["g_x2stretch"])
        assert list(L.prob_domain)[:bi] == list(lv["prob_domain"])
        boxes = np.ctypeslib.as_array(L.boxes, shape=(L.n_boxes, bi)).copy()
        offs = np.ctypeslib.as_array(L.box_offsets, shape=(L.n_boxes,)).copy()
        data = np.ctypeslib.as_array(f.data, shape=(off + L.data_len,))[off:].copy()
        assert np.array_equal(boxes, lv["boxes"]) and np.array_equal(offs, lv["box_offsets"]) and np.array_equal(data, lv["data"])
        back["levels"].append(dict(lv, boxes=boxes, box_offsets=offs, data=data))
        off += L.data_len
    cfg = oracle.make_config(dims, geom, 0)
    slab = dict(r_inj=1e12, ph_inj_switch=1, min_r=0, max_r=0, min_theta=0, max_theta=0, fps=5.0)
    a, _ = oracle.hydro_ingest(cfg, back, slab)
    c, _ = oracle.hydro_ingest(cfg, raw, slab)
    assert a["num_elements"] == c["num_elements"] > 100 and np.array_equal(a["r0"], c["r0"]) and np.array_equal(a["dens"], c["dens"])
    h5.mcrat_host_free_chombo(C.byref(hc))
    assert h5.mcrat_host_read_chombo(b"/nonexistent", int(three), 1.0, 1.0, 1.0, C.byref(hc)) == -1
    if not three:                       # a 2-D file has no lo_k / hi_k members and no domBeg3: it cannot be read as 3-D
        assert h5.mcrat_host_read_chombo(path, 1, 1.0, 1.0, 1.0, C.byref(hc)) == -2


def test_writers_and_readers_in_several_threads(h5, tmp_path):
    """mcrat_host_run_ranks' asynchronous output has print_photons (writer thread, frame F) and get_hydro -> mcrat_host_read_flash (calling thread,
    frame F + 1) in HDF5 at the same time, two of each with two pools per process; the default HDF5 build is not thread-safe, so the library
    serialises its own HDF5 entry points on one lock (mcrat_hip_host_h5.c).  Four threads hammer them here on real files; every call must succeed
    and every file must hold what was written (without the lock this crashes or corrupts HDF5's ID tables within a few hundred calls)."""
    import threading
    e = h5.engine
    raw = synth.flash_raw_blocks(2e9, 8, 16, 8, 1e12 - 1.6e10, seed=3)
    keep = []

    def ptr(a, dtype=np.float64, ctype=C.c_double):
        arr = np.ascontiguousarray(a, dtype=dtype)
        keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(ctype))
    n = len(raw["node_type"])
    b = e.FlashBlocks(n, 3, 2, ptr(raw["coordinates"]), ptr(raw["block_size"]), ptr(raw["node_type"], np.int32, C.c_int), ptr(raw["velx"]),
                      ptr(raw["vely"]), ptr(raw["dens"]), ptr(raw["pres"]), 1.0, 1.0, 1.0)
    flash = str(tmp_path / "frame_0001").encode()
    assert h5.writer.fixture_write_flash(flash, C.byref(b)) == 0
    h5.mcrat_host_print_photon_arrays.restype = C.c_int
    h5.mcrat_host_print_photon_arrays.argtypes = [C.POINTER(e.OutputColumns), C.c_int, C.c_char_p, C.c_int, C.c_void_p]
    h5.mcrat_host_h5_read.restype = C.c_int
    h5.mcrat_host_h5_read.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    h5.mcrat_host_h5_threadsafe.restype = C.c_int
    assert h5.mcrat_host_h5_threadsafe() in (0, 1)
    h5.mcrat_host_h5_lock()                                 # the lock is recursive and exported for callers' own HDF5 calls
    h5.mcrat_host_h5_lock()
    h5.mcrat_host_h5_unlock()
    h5.mcrat_host_h5_unlock()
    frames, m = 40, 257
    errors = []

    def writer(rank):
        rng = np.random.default_rng(rank)
        for f in range(frames):
            cols = e.OutputColumns()
            cols.count = m
            arrs = {}
            for k, name in enumerate(e.OUTPUT_COLUMNS):
                arrs[name] = np.full(m, 1000.0 * rank + f + 0.001 * k) + rng.random(m) * 0.0
                setattr(cols, name, arrs[name].ctypes.data_as(C.POINTER(C.c_double)))
            cols.type = None
            rc = h5.mcrat_host_print_photon_arrays(C.byref(cols), f, (str(tmp_path) + "/").encode(), rank, None)
            if rc:
                errors.append(("print", rank, f, rc))

    def reader(k):
        for f in range(frames):
            hf = h5.HostFlash()
            rc = h5.mcrat_host_read_flash(flash, 1.0, 1.0, 1.0, C.byref(hf))
            if rc or hf.blocks.n_blocks != n or not np.array_equal(np.ctypeslib.as_array(hf.blocks.dens, shape=(n, 1, 8, 8)), raw["dens"]):
                errors.append(("read", k, f, rc))
            h5.mcrat_host_free_flash(C.byref(hf))
    threads = [threading.Thread(target=writer, args=(r,)) for r in (0, 1)] + [threading.Thread(target=reader, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    for rank in (0, 1):
        for f in (0, frames - 1):
            out = np.zeros(m)
            got = C.c_int(0)
            assert h5.mcrat_host_h5_read((str(tmp_path) + "/mc_proc_%d.h5" % rank).encode(), str(f).encode(), b"P0", 0, out.ctypes.data, m, C.byref(got)) == 0
            assert got.value == m and np.all(out == 1000.0 * rank + f)
