"""mc_proc_<rank>.h5 against the reference's own format, not against this repository's reader: the files the host
writer produces (mcrat_host_print_photon_arrays, the HDF5 half of printPhotons) are dumped with HDF5's own `h5dump -H -p` and compared
with tests/golden/mc_proc_listing.json -- the dataset names, memory types, rank, chunking, maximum dimensions and append behaviour read
off the reference's HDF5 call sequence (Src/mcrat_io.c:114-836) by tests/golden/make_h5_listing.py -- for every combination of the
COMV_SWITCH / STOKES_SWITCH / SAVE_TYPE switches.  No GPU needed: the arrays are the test's."""
import ctypes as C
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LISTING = json.load(open(os.path.join(HERE, "golden", "mc_proc_listing.json")))
H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
NATIVE = {"H5T_NATIVE_DOUBLE": "H5T_IEEE_F64LE", "H5T_NATIVE_CHAR": "H5T_STD_I8LE"}      # x86-64 little endian
COLUMN = {"P0": "p0", "P1": "p1", "P2": "p2", "P3": "p3", "COMV_P0": "comv_p0", "COMV_P1": "comv_p1", "COMV_P2": "comv_p2", "COMV_P3": "comv_p3",
          "R0": "r0", "R1": "r1", "R2": "r2", "S0": "s0", "S1": "s1", "S2": "s2", "S3": "s3", "NS": "num_scatt", "PW": "weight", "PT": "type"}


@pytest.fixture(scope="module")
def h5():
    from mcrat_amd.host import binding
    lib = binding.host_h5()
    if lib is None or not os.path.exists(H5DUMP):
        pytest.skip("no HDF5 C library / h5dump in this image")
    return lib


def _columns(engine, n, seed, comv, stokes, save_type):
    rng = np.random.default_rng(seed)
    o, keep = engine.OutputColumns(), {}
    o.count = n
    for f in engine.OUTPUT_COLUMNS:
        on = not ((f.startswith("comv") and not comv) or (f in ("s0", "s1", "s2", "s3") and not stokes))
        keep[f] = rng.normal(size=n) if on else None
        setattr(o, f, keep[f].ctypes.data_as(C.POINTER(C.c_double)) if on else None)
    keep["type"] = np.frombuffer(rng.choice(list(b"ikc"), n).astype(np.uint8).tobytes(), dtype="S1").copy() if save_type else None
    o.type = keep["type"].ctypes.data_as(C.c_char_p) if save_type else None
    return o, keep


def _parse(dump):
    """h5dump -H -p -> {dataset: (datatype, current dims, maximum dims, chunk)} per group"""
    out = {}
    for g in re.finditer(r'GROUP "(\d+)" \{(.*?)\n   \}', dump, re.S):
        sets = {}
        for d in re.finditer(r'DATASET "(\w+)" \{\s*DATATYPE\s+(\w+)\s*DATASPACE\s+SIMPLE \{ \( (\d+) \) / \( (\w+) \) \}.*?CHUNKED \( (\d+) \)', g.group(2), re.S):
            sets[d.group(1)] = (d.group(2), int(d.group(3)), d.group(4), int(d.group(5)))
        out[g.group(1)] = sets
    return out


@pytest.mark.parametrize("comv,stokes,save_type", [(c, s, t) for c in (0, 1) for s in (0, 1) for t in (0, 1)])
def test_layout_is_the_reference_call_sequence(h5, tmp_path, comv, stokes, save_type):
    from mcrat_amd import engine
    d = str(tmp_path) + "/"
    on = {"COMV_SWITCH": comv, "STOKES_SWITCH": stokes, "SAVE_TYPE": save_type, None: 1}
    expected = [ds for ds in LISTING["datasets"] if on[ds["switch"]]]
    # frame 7: written once; frame 8: written by two injection batches of the same rank (the group exists the second time:
    # the datasets are extended by the new count and written at the old end, mcrat_io.c:402-424)
    a, ka = _columns(engine, 37, 1, comv, stokes, save_type)
    b, kb = _columns(engine, 50, 2, comv, stokes, save_type)
    c, kc = _columns(engine, 21, 3, comv, stokes, save_type)
    assert h5.mcrat_host_print_photon_arrays(C.byref(a), 7, d.encode(), 3, None) == 0
    assert h5.mcrat_host_print_photon_arrays(C.byref(b), 8, d.encode(), 3, None) == 0
    assert h5.mcrat_host_print_photon_arrays(C.byref(c), 8, d.encode(), 3, None) == 0
    name = LISTING["file_name"]
    path = d + name["prefix"] + "3" + name["suffix"]                       # "%s%s%d%s": dir, "mc_proc_", angle_rank, ".h5"
    assert os.path.exists(path)
    dump = subprocess.run([H5DUMP, "-H", "-p", path], capture_output=True, text=True, check=True).stdout
    groups = _parse(dump)
    assert set(groups) == {"7", "8"}                                        # "%d" of the frame
    decl = LISTING["declarations"]
    assert decl["rank"] == 1 and decl["maxdims"] == "H5S_UNLIMITED" and decl["dims"] == decl["dims_weight"] == "net_num_ph"
    for grp, first, total in (("7", 37, 37), ("8", 50, 71)):
        sets = groups[grp]
        assert set(sets) == {ds["name"] for ds in expected}                 # exactly the reference's datasets for these switches
        for ds in expected:
            dtype, cur, mx, chunk = sets[ds["name"]]
            assert dtype == NATIVE[ds["memory_type"]], ds["name"]
            # one dimension; the dataspace of the first write, unlimited; chunk = the first write's photon count (H5Pset_chunk(prop, rank, dims))
            assert LISTING["dataspaces"][ds["dataspace"]]["maxdims_var"] == "maxdims" and LISTING["chunking"][ds["dcpl"]]["dims_var"] in ("dims", "dims_weight")
            assert (cur, mx, chunk) == (total, "H5S_UNLIMITED", first), (grp, ds["name"])
    # append behaviour: size = old + new, the new values at offset old
    ap = LISTING["append"]
    assert ap["size_is_old_plus_new"] and ap["hyperslab_offset_is_old"] and ap["hyperslab"] == "H5S_SELECT_SET" and ap["extends"] >= len(LISTING["datasets"])
    for ds in expected:
        is_char = ds["memory_type"] == "H5T_NATIVE_CHAR"
        for grp, parts in (("7", [ka]), ("8", [kb, kc])):
            want = np.concatenate([p[COLUMN[ds["name"]]] for p in parts])
            buf = np.empty(len(want), dtype="S1" if is_char else np.float64)
            n = C.c_int()
            assert h5.mcrat_host_h5_read(path.encode(), grp.encode(), ds["name"].encode(), int(is_char), buf.ctypes.data, len(buf), C.byref(n)) == 0
            assert n.value == len(want) and np.array_equal(buf, want), (grp, ds["name"])


def test_listing_is_current_with_the_reference_where_it_is_present():
    """in this container the committed listing must be what the script extracts from /root/reference today"""
    if not os.path.exists("/root/reference/Src/mcrat_io.c"):
        pytest.skip("no reference here (GPU box)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_h5_listing", os.path.join(HERE, "golden", "make_h5_listing.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    before = open(os.path.join(HERE, "golden", "mc_proc_listing.json")).read()
    m.main()
    assert open(os.path.join(HERE, "golden", "mc_proc_listing.json")).read() == before
    names = [d["name"] for d in LISTING["datasets"]]
    assert names == ["P0", "P1", "P2", "P3", "COMV_P0", "COMV_P1", "COMV_P2", "COMV_P3", "R0", "R1", "R2", "S0", "S1", "S2", "S3", "PT", "NS", "PW"]
