"""CPU checks of the output row, host side: mc_chkpt_<rank>.dat in the reference's byte layout
(saveCheckpoint, Src/mcrat_io.c:838-1009) and its reader (readCheckpoint, :1011-1134)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from mcrat_amd import synth


class PhotonList(C.Structure):
    _fields_ = [("photons", C.c_void_p), ("sorted_indexes", C.c_void_p), ("num_photons", C.c_int), ("num_null_photons", C.c_int),
                ("list_capacity", C.c_int)]


@pytest.fixture(scope="module")
def host():
    from mcrat_amd import build, engine
    from mcrat_amd.host import build_host
    build.build()
    lib = C.CDLL(build_host.build())
    lib.mcrat_host_save_checkpoint.restype = C.c_int
    lib.mcrat_host_save_checkpoint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(PhotonList), C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mcrat_host_read_checkpoint.restype = C.c_int
    lib.mcrat_host_read_checkpoint.argtypes = [C.c_char_p, C.POINTER(PhotonList), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    lib.engine = engine
    return lib


def _records(engine, n=300, seed=5):
    frame, ph, cfg = synth.config2(n_photons=n, nzc=4, seed=seed)
    aos = synth.photons_to_aos(ph, engine.PHOTON_DTYPE)
    aos["type"][::17] = b"N"
    aos["weight"][::17] = 0
    return aos


def _save(host, d, aos, frame, frame2, scatt_frame, t, last_frame, rank=3, size=8):
    l = PhotonList(aos.ctypes.data, None, len(aos), 0, len(aos))
    return host.mcrat_host_save_checkpoint((str(d) + "/").encode(), frame, frame2, scatt_frame, t, None, C.byref(l), len(aos), last_frame, rank, size, 0)


def test_checkpoint_bytes_are_the_references_layout(host, tmp_path):
    aos = _records(host.engine)
    assert aos.dtype.itemsize == 176
    path = tmp_path / "mc_chkpt_3.dat"
    # a frame in the middle of a run: CONTINUE header, all records (mcrat_io.c:846-896)
    assert _save(host, tmp_path, aos, frame=200, frame2=203, scatt_frame=257, t=51.4, last_frame=3000) == 0
    raw = path.read_bytes()
    head = struct.pack("=i", 8) + b"c" + struct.pack("=iii", 200, 203, 257) + struct.pack("=d", 51.4) + struct.pack("=i", len(aos))
    assert raw[:len(head)] == head and len(raw) == len(head) + 176 * len(aos)
    assert raw[len(head):] == aos.tobytes()
    # the next save keeps the previous file as _old (:849)
    assert _save(host, tmp_path, aos, 200, 203, 258, 51.6, 3000) == 0
    assert (tmp_path / "mc_chkpt_3.dat_old").read_bytes() == raw
    # the injection frame itself (scatt_frame == frame): the old file is removed, not kept (:898-901)
    os.remove(tmp_path / "mc_chkpt_3.dat_old")
    assert _save(host, tmp_path, aos, 200, 203, 200, 40.0, 3000) == 0
    assert not (tmp_path / "mc_chkpt_3.dat_old").exists()
    assert path.read_bytes()[4:5] == b"c"
    # after the last hydro frame: INITALIZE header without scatt_frame / time / count, records still follow (:947-979)
    assert _save(host, tmp_path, aos, 200, 203, 3000, 600.0, 3000) == 0
    raw = path.read_bytes()
    head = struct.pack("=i", 8) + b"i" + struct.pack("=ii", 200, 203)
    assert raw[:len(head)] == head and raw[len(head):] == aos.tobytes()
    # an unwritable directory is reported like the reference does (return 1)
    assert _save(host, tmp_path / "missing", aos, 1, 2, 3, 0.0, 10) == 1


def test_checkpoint_round_trip_with_the_references_restart_conventions(host, tmp_path):
    aos = _records(host.engine, n=257)
    assert _save(host, tmp_path, aos, 200, 203, 257, 51.4, 3000) == 0
    l = PhotonList()
    frame2, framestart, scatt, t, size = C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0), C.c_int(0)
    restart = C.create_string_buffer(2)
    assert host.mcrat_host_read_checkpoint((str(tmp_path) + "/").encode(), C.byref(l), C.byref(frame2), C.byref(framestart), C.byref(scatt),
                                           restart, C.byref(t), 3, C.byref(size)) == 0
    assert (restart.raw[:1], framestart.value, frame2.value, scatt.value, t.value, size.value) == (b"c", 200, 203, 258, 51.4, 8)   # scatt_frame + 1
    assert (l.list_capacity, l.num_null_photons, l.num_photons) == (257, int((aos["type"] == b"N").sum()), 257 - int((aos["type"] == b"N").sum()))
    back = np.frombuffer((C.c_char * (176 * 257)).from_address(l.photons), dtype=aos.dtype).copy()
    for k in aos.dtype.names:
        if k in ("recalc_properties", "time_to_scatter", "total_optical_depth"):      # not carried over by readCheckpoint (:1064-1083)
            continue
        assert np.array_equal(back[k], aos[k]), k
    assert (back["recalc_properties"] == 1).all() and (back["time_to_scatter"] == 0).all()
    C.CDLL(None).free(C.c_void_p(l.photons))
    # the INITALIZE file: framestart + 1, scatt_framestart = framestart, no photons (:1107-1121)
    assert _save(host, tmp_path, aos, 200, 203, 3000, 600.0, 3000) == 0
    l = PhotonList()
    assert host.mcrat_host_read_checkpoint((str(tmp_path) + "/").encode(), C.byref(l), C.byref(frame2), C.byref(framestart), C.byref(scatt),
                                           restart, C.byref(t), 3, C.byref(size)) == 0
    assert (restart.raw[:1], framestart.value, scatt.value, l.photons) == (b"i", 201, 201, None)
    # no file: defaults (:1127-1131)
    framestart.value = 77
    assert host.mcrat_host_read_checkpoint((str(tmp_path) + "/").encode(), C.byref(l), C.byref(frame2), C.byref(framestart), C.byref(scatt),
                                           restart, C.byref(t), 5, C.byref(size)) == 0
    assert (restart.raw[:1], scatt.value) == (b"i", 77)
    # a truncated file is an error, not a short list
    raw = (tmp_path / "mc_chkpt_3.dat").read_bytes()
    assert _save(host, tmp_path, aos, 200, 203, 257, 51.4, 3000) == 0
    full = (tmp_path / "mc_chkpt_3.dat").read_bytes()
    (tmp_path / "mc_chkpt_3.dat").write_bytes(full[:-100])
    assert host.mcrat_host_read_checkpoint((str(tmp_path) + "/").encode(), C.byref(l), C.byref(frame2), C.byref(framestart), C.byref(scatt),
                                           restart, C.byref(t), 3, C.byref(size)) == -2


def test_output_floor_writes_renames_and_times(host, tmp_path):
    """mcrat_host_output_floor: saveCheckpoint's file sequence (rename to _old, fopen "wb", fwrite, fclose; mcrat_io.c:846-900) on payloads of
    zeros, shared by threads -- what bench.py holds the asynchronous writer of mcrat_host_run_ranks against"""
    host.mcrat_host_output_floor.restype = C.c_int
    host.mcrat_host_output_floor.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
    ms = C.c_double(-1)
    d = (str(tmp_path) + "/").encode()
    assert host.mcrat_host_output_floor(d, 7, 1000, 3, 3, C.byref(ms)) == 0
    assert ms.value > 0
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["floor_%d.dat" % r for r in range(7)] + ["floor_%d.dat_old" % r for r in range(7)])
    assert all(os.path.getsize(tmp_path / n) == 1000 for n in names)
    assert host.mcrat_host_output_floor(d, 0, 1000, 3, 3, C.byref(ms)) != 0
    assert host.mcrat_host_output_floor((str(tmp_path) + "/missing/").encode(), 2, 10, 1, 1, C.byref(ms)) == 1
