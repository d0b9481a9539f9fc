"""GPU tests of the output row: what printPhotons and saveCheckpoint take from the photon list after a
frame -- the compacted weight != 0 columns (mcrat_io.c:137-181), struct photon records in pieces, the checkpoint file
streamed from the device, and the per-frame HDF5 datasets of mc_proc_<rank>.h5."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


class PhotonList(C.Structure):
    _fields_ = [("photons", C.c_void_p), ("sorted_indexes", C.c_void_p), ("num_photons", C.c_int), ("num_null_photons", C.c_int),
                ("list_capacity", C.c_int)]


def _engine_after_a_frame(hip, n=20000, stokes=1, passes=300):
    frame, ph, cfg = synth.config2(n_photons=n, nzc=8, stokes=stokes, lumi=1e53)
    rng = np.random.default_rng(2)
    dead = rng.random(n) < 0.15                      # absorbed / NULL slots: weight 0 (photons.c:181-214)
    ph["weight"][dead] = 0.0
    ph["type"][dead & (rng.random(n) < 0.5)] = b"N"
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(3, 0.0, 0.2)
    st = e.run(passes)
    assert st.frame_scatt_cnt > 20
    return e, dead


def test_output_columns_are_the_weighted_photons_in_slot_order(hip):
    e, dead = _engine_after_a_frame(hip)
    full = e.get_photons()
    out = e.get_output()
    keep = np.asarray(full["weight"]) != 0
    assert np.array_equal(keep, ~dead) and len(out["p0"]) == int(keep.sum())
    for k in hip.OUTPUT_COLUMNS:
        assert np.array_equal(out[k], np.asarray(full[k])[keep], equal_nan=True), k
    assert np.array_equal(out["type"], np.asarray(full["type"])[keep])
    # capacity too small: refused with the needed count, nothing written
    o = hip.OutputColumns()
    small = np.zeros(10)
    o.count, o.p0 = 10, small.ctypes.data_as(C.POINTER(C.c_double))
    assert e.lib.mcrat_hip_get_output(e.ctx, C.byref(o)) == -1 and o.count == int(keep.sum()) and not small.any()
    # an all-dead list has nothing to write
    full["weight"][:] = 0.0
    e.set_photons(full)
    assert len(e.get_output()["p0"]) == 0
    e.close()


def test_records_in_pieces_equal_the_whole_list(hip):
    e, _ = _engine_after_a_frame(hip, n=5000)
    whole = e.get_photons_aos()
    for first, count in ((0, 5000), (0, 1), (4999, 1), (1234, 2049), (256, 512)):
        piece = e.get_photons_range(first, count)
        for k in whole.dtype.names:
            assert np.array_equal(piece[k], whole[k][first:first + count], equal_nan=piece[k].dtype.kind == "f"), (first, count, k)
        raw = np.frombuffer(piece.tobytes(), dtype=np.uint8).reshape(count, 176)
        assert not raw[:, 1:8].any() and not raw[:, 140:144].any() and not raw[:, 156:160].any()      # the padding bytes are zero
    with pytest.raises(hip.McratHipError):
        e.get_photons_range(4000, 1001)
    e.close()


def test_checkpoint_streamed_from_the_device_equals_the_one_written_from_host_records(hip, tmp_path):
    from mcrat_amd.host import build_host
    host = C.CDLL(build_host.build())
    host.mcrat_host_save_checkpoint.restype = C.c_int
    host.mcrat_host_save_checkpoint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(PhotonList), C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int]
    e, _ = _engine_after_a_frame(hip, n=7000)
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    assert host.mcrat_host_save_checkpoint((str(tmp_path / "a") + "/").encode(), 200, 203, 257, 51.4, e.ctx, None, 7000, 3000, 0, 4, 0) == 0
    aos = e.get_photons_range(0, 7000)
    l = PhotonList(aos.ctypes.data, None, 7000, 0, 7000)
    assert host.mcrat_host_save_checkpoint((str(tmp_path / "b") + "/").encode(), 200, 203, 257, 51.4, None, C.byref(l), 7000, 3000, 0, 4, 0) == 0
    a, b = (tmp_path / "a" / "mc_chkpt_0.dat").read_bytes(), (tmp_path / "b" / "mc_chkpt_0.dat").read_bytes()
    assert a == b and len(a) == 4 + 1 + 12 + 8 + 4 + 176 * 7000
    assert a[:5] == struct.pack("=i", 4) + b"c"
    e.close()


def test_checkpoint_with_the_cyclosynchrotron_switch_converts_the_resident_list(hip, tmp_path):
    """saveCheckpoint converts 'k' photons with weight != 0 to 'c' IN the list (mcrat_io.c:896-900), so the next frame's phAbsCyclosynch and the PT
    column see the converted types: mcrat_host_save_checkpoint(..., ctx, ..., cyclosynchrotron_switch = 1) does it on the device list, and the file
    holds the converted records -- the same as converting host records and writing those"""
    from mcrat_amd.host import build_host
    host = C.CDLL(build_host.build())
    host.mcrat_host_save_checkpoint.restype = C.c_int
    host.mcrat_host_save_checkpoint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(PhotonList), C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int]
    frame, ph, cfg = synth.config2(n_photons=3000, nzc=8, stokes=1, lumi=1e53)
    aos = synth.photons_to_aos(ph, hip.PHOTON_DTYPE)
    aos["type"][::3] = b"k"
    aos["weight"][::6] = 0.0                                     # a 'k' photon with weight 0 stays 'k'
    aos["type"][1::7] = b"p"
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons_aos(aos)
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    assert host.mcrat_host_save_checkpoint((str(tmp_path / "a") + "/").encode(), 200, 203, 257, 51.4, e.ctx, None, 3000, 3000, 0, 4, 1) == 0
    got = e.get_photons_aos()
    want = aos.copy()
    conv = (want["type"] == b"k") & (want["weight"] != 0)
    assert conv.sum() > 400 and ((want["type"] == b"k") & (want["weight"] == 0)).sum() > 100
    want["type"][conv] = b"c"
    assert np.array_equal(got["type"], want["type"])             # the resident list has changed
    l = PhotonList(aos.ctypes.data, None, 3000, 0, 3000)          # the host path converts its records in place too
    assert host.mcrat_host_save_checkpoint((str(tmp_path / "b") + "/").encode(), 200, 203, 257, 51.4, None, C.byref(l), 3000, 3000, 0, 4, 1) == 0
    assert np.array_equal(aos["type"], want["type"])
    a, b = (tmp_path / "a" / "mc_chkpt_0.dat").read_bytes(), (tmp_path / "b" / "mc_chkpt_0.dat").read_bytes()
    head = 4 + 1 + 12 + 8 + 4
    ra, rb = np.frombuffer(a[head:], dtype=hip.PHOTON_DTYPE), np.frombuffer(b[head:], dtype=hip.PHOTON_DTYPE)
    assert np.array_equal(ra["type"], want["type"]) and np.array_equal(rb["type"], want["type"])
    e.close()


def test_hdf5_frame_datasets_like_printPhotons(hip, tmp_path):
    from mcrat_amd.host import build_host
    path = build_host.build_h5()
    if path is None:
        pytest.skip("no HDF5 C library in this image")
    h5 = C.CDLL(path)
    h5.mcrat_host_print_photons.restype = C.c_int
    h5.mcrat_host_print_photons.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    h5.mcrat_host_h5_read.restype = C.c_int
    h5.mcrat_host_h5_read.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    e, _ = _engine_after_a_frame(hip, n=6000)
    out = e.get_output()
    m = len(out["p0"])
    d = (str(tmp_path) + "/").encode()
    assert h5.mcrat_host_print_photons(e.ctx, 257, d, 2, 1, 1, 1, None) == 0
    file = os.path.join(str(tmp_path), "mc_proc_2.h5").encode()

    def read(name, is_char=False):
        n = C.c_int(0)
        assert h5.mcrat_host_h5_read(file, b"257", name.encode(), int(is_char), None, 0, C.byref(n)) == 0, name
        a = np.empty(n.value, dtype="S1" if is_char else np.float64)
        assert h5.mcrat_host_h5_read(file, b"257", name.encode(), int(is_char), a.ctypes.data, n.value, C.byref(n)) == 0
        return a
    names = dict(P0="p0", P1="p1", P2="p2", P3="p3", COMV_P0="comv_p0", COMV_P1="comv_p1", COMV_P2="comv_p2", COMV_P3="comv_p3",
                 R0="r0", R1="r1", R2="r2", S0="s0", S1="s1", S2="s2", S3="s3", NS="num_scatt", PW="weight")
    for dset, col in names.items():
        assert np.array_equal(read(dset), out[col], equal_nan=True), dset
    assert np.array_equal(read("PT", True), out["type"])
    # the same frame printed again (the reference does that when photons are injected into an existing frame group): extended
    assert h5.mcrat_host_print_photons(e.ctx, 257, d, 2, 1, 1, 1, None) == 0
    again = read("R1")
    assert len(again) == 2 * m and np.array_equal(again[:m], out["r1"]) and np.array_equal(again[m:], out["r1"])
    # another frame -> another group; COMV / STOKES / SAVE_TYPE OFF builds do not create those datasets
    assert h5.mcrat_host_print_photons(e.ctx, 258, d, 2, 0, 0, 0, None) == 0
    n = C.c_int(0)
    assert h5.mcrat_host_h5_read(file, b"258", b"P0", 0, None, 0, C.byref(n)) == 0 and n.value == m
    for absent in (b"COMV_P0", b"S0", b"PT"):
        assert h5.mcrat_host_h5_read(file, b"258", absent, 0, None, 0, C.byref(n)) != 0
    e.close()


def test_device_streamed_checkpoint_headers_in_the_three_cases_of_saveCheckpoint(hip, tmp_path):
    """the file streamed from the device (ctx branch) against bytes assembled here from the reference's write sequence
    (Src/mcrat_io.c:871-903 CONTINUE in the middle of a run, :918-961 the injection frame itself, :963-1000 after the last hydro frame:
    INITALIZE header without scatt_frame / time / count) -- not against the host-list writer"""
    from mcrat_amd.host import binding as B
    host = B.host()
    e, _ = _engine_after_a_frame(hip, n=3000)
    rec = e.get_photons_range(0, 3000)
    d = str(tmp_path) + "/"
    path, old = tmp_path / "mc_chkpt_5.dat", tmp_path / "mc_chkpt_5.dat_old"
    body = rec.tobytes()
    assert len(body) == 176 * 3000

    def save(frame, frame2, scatt_frame, t, last):
        return host.mcrat_host_save_checkpoint(d.encode(), frame, frame2, scatt_frame, t, e.ctx, None, 3000, last, 5, 16, 0)
    # (1) scatt_frame != last_frame, != frame: angle_size, 'c', frame, frame2, scatt_frame, time_now, ph_num, records
    assert save(200, 203, 257, 51.4, 3000) == 0
    first = path.read_bytes()
    assert first == struct.pack("=i", 16) + b"c" + struct.pack("=iii", 200, 203, 257) + struct.pack("=d", 51.4) + struct.pack("=i", 3000) + body
    assert not old.exists()                                                  # "cp" of a file that did not exist yet
    assert save(200, 203, 258, 51.6, 3000) == 0
    assert old.read_bytes() == first and path.read_bytes()[5:17] == struct.pack("=iii", 200, 203, 258)
    # (2) scatt_frame == frame: the previous file is removed ("rm"), no _old is made; same CONTINUE header
    os.remove(old)
    assert save(300, 303, 300, 60.0, 3000) == 0
    assert not old.exists()
    assert path.read_bytes() == struct.pack("=i", 16) + b"c" + struct.pack("=iii", 300, 303, 300) + struct.pack("=d", 60.0) + struct.pack("=i", 3000) + body
    # (3) scatt_frame == last_frame: 'i', frame, frame2, then the records with no count in front; the previous file kept as _old
    before = path.read_bytes()
    assert save(300, 303, 3000, 600.0, 3000) == 0
    assert old.read_bytes() == before
    assert path.read_bytes() == struct.pack("=i", 16) + b"i" + struct.pack("=ii", 300, 303) + body
    e.close()


def test_outbox_holds_the_frame_while_the_photons_move_on(hip):
    """mcrat_hip_outbox_post / _wait: the records and output columns of the moment of the post, although the loop has gone on in between
    (what mcrat_host_run_ranks' writer thread reads while the next frame propagates); posting twice reuses the buffers"""
    e, dead = _engine_after_a_frame(hip, n=7000)
    want_rec, want_out = e.get_photons_range(0, 7000), e.get_output()
    box = e.outbox_create()
    e.outbox_post(box, True, True)
    st = e.run(200)                                              # the photons change
    assert st.iterations > 0 and not np.array_equal(e.get_photons_range(0, 7000)["r0"], want_rec["r0"])
    rec, out = e.outbox_wait(box)
    assert rec.tobytes() == want_rec.tobytes()
    for k in hip.OUTPUT_COLUMNS:
        assert np.array_equal(out[k], want_out[k], equal_nan=True), k
    assert np.array_equal(out["type"], want_out["type"])
    # records only, then columns only
    now_rec, now_out = e.get_photons_range(0, 7000), e.get_output()
    e.outbox_post(box, True, False)
    rec, out = e.outbox_wait(box)
    assert rec.tobytes() == now_rec.tobytes() and len(out["p0"]) == 0
    e.outbox_post(box, False, True)
    rec, out = e.outbox_wait(box)
    assert rec is None and np.array_equal(out["weight"], now_out["weight"])
    with pytest.raises(Exception):
        e.outbox_post(box, False, False)
    e.outbox_destroy(box)
    e.close()
