"""The engine against facts read off the reference's text (tests/golden/ref_facts.json, made by tests/golden/make_ref_facts.py from
/root/reference/Src: the constants of mclib.c:4-5, struct photon / photonList of mcrat.h:142-180, the table grid of hot_x_section.h:2-10,
saveCheckpoint's fwrite order of mcrat_io.c:871-903, readMcPar's read order of mcrat_io.c:1136-1237).  The reference cannot be built or run here
(GSL is absent) and ships no vectors; what its text fixes -- values, orders, layouts -- is the one kind of reference-held evidence this image
allows, and this file uses all of it that is not already pinned by tests/test_h5_layout.py.  CPU only."""
import ctypes as C
import json
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTS = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_facts.json")))


def _defs(path, pattern):
    return {m.group(1): m.group(2) for m in re.finditer(pattern, open(os.path.join(ROOT, path)).read())}


def test_the_ten_constants_are_the_references_literals():
    ref = {k: float(v["literal"]) for k, v in FACTS["constants"].items()}
    assert len(ref) == 10
    dev = {k: float(v) for k, v in _defs("mcrat_amd/csrc/device_types.hpp", r"constexpr double (\w+) = ([0-9.eE+-]+);").items()}
    for name in ("A_RAD", "C_LIGHT", "PL_CONST", "K_B", "M_P", "THOM_X_SECT", "M_EL"):
        assert dev[name] == ref[name], name                                   # bit for bit: the same decimal literal gives the same double
    orc = {k: float(v) for k, v in _defs("oracle/mcrat_oracle.h", r"#define ORC_(\w+)\s+([0-9.eE+-]+)").items()}
    for name, v in orc.items():
        if name in ref:
            assert v == ref[name], name
    assert {"A_RAD", "C_LIGHT", "PL_CONST", "K_B", "M_P", "THOM_X_SECT", "M_EL"} <= set(orc)
    # the charge (cyclo-synchrotron: nu_c = e B / 2 pi m c) wherever the device code names it
    for path in ("mcrat_amd/csrc/cs_device.hpp", "mcrat_amd/csrc/staging.hip"):
        assert float(_defs(path, r"constexpr double (CHARGE_EL) = ([0-9.eE+-]+);")["CHARGE_EL"]) == ref["CHARGE_EL"], path
    # every other appearance of one of the ten names as a constexpr / #define in the product or the oracle carries the reference's value
    for dirpath, _, files in list(os.walk(os.path.join(ROOT, "mcrat_amd", "csrc"))) + list(os.walk(os.path.join(ROOT, "oracle"))):
        for f in files:
            if f.endswith((".hpp", ".hip", ".h", ".c")):
                for name, val in re.findall(r"(?:constexpr double|#define)\s+(?:ORC_)?(\w+)\s*=?\s*([0-9]+\.[0-9]+e[+-]?[0-9]+)\b", open(os.path.join(dirpath, f)).read()):
                    if name in ref:
                        assert float(val) == ref[name], (f, name)


def test_struct_photon_and_photon_list_are_the_references_member_for_member():
    from mcrat_amd import engine
    members = FACTS["struct_photon"]["members"]
    dt = engine.PHOTON_DTYPE
    assert [m["name"] for m in members] == [n for n in dt.names if not n.startswith("_")] or [m["name"] for m in members] == list(dt.names)
    ctype = {"char": (1, "S1"), "double": (8, "f8"), "int": (4, "i4")}
    # the x86-64 layout of the reference's declaration order: every member aligned to its own size, the record to 8 (mcrat.h:142-171)
    off = 0
    for m in members:
        size, kind = ctype[m["type"]]
        off = (off + size - 1) // size * size
        assert dt.fields[m["name"]][1] == off, (m["name"], dt.fields[m["name"]][1], off)
        assert dt.fields[m["name"]][0].itemsize == size and dt.fields[m["name"]][0].kind in ("S", "f", "i"), m["name"]
        off += size
    assert (off + 7) // 8 * 8 == dt.itemsize == 176
    # the C ABI header declares the same members in the same order (include/mcrat_hip.h: mcrat_hip_photon, mcrat_hip_photon_list)
    hdr = open(os.path.join(ROOT, "include", "mcrat_hip.h")).read()
    body = re.search(r"typedef struct mcrat_hip_photon \{(.*?)\} mcrat_hip_photon;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    got = []
    for line in body.split(";"):
        d = re.match(r"\s*(char|double|int)\s+(.*)", line.strip(), flags=re.S)
        if d:
            got += [(d.group(1), n.strip()) for n in d.group(2).split(",")]
    assert got == [(m["type"], m["name"]) for m in members]
    body = re.search(r"typedef struct mcrat_hip_photon_list \{(.*?)\} mcrat_hip_photon_list;", hdr, flags=re.S).group(1)
    names = re.findall(r"\b(\w+);", body)
    assert names == [m["name"] for m in FACTS["struct_photon_list"]["members"]]
    kinds = [("*" in line) for line in body.strip().split("\n") if ";" in line]
    assert kinds == [m["type"].endswith("*") for m in FACTS["struct_photon_list"]["members"]]


def test_the_cross_section_table_grid():
    g = FACTS["hot_table_grid"]
    from mcrat_amd import engine
    import inspect
    # the defaults of the binding and of the host reader are the reference's grid (221 x 81 nodes on [-12, 6] x [-4, 4])
    sig = inspect.signature(engine.Engine.create_hot_cross_section)
    assert sig.parameters["n_ph_e"].default == g["N_PH_E"] and sig.parameters["n_t"].default == g["N_T"]
    assert tuple(sig.parameters["grid"].default) == (g["LOG_PH_E_MIN"], g["LOG_PH_E_MAX"], g["LOG_T_MIN"], g["LOG_T_MAX"])
    assert tuple(inspect.signature(engine.Engine.set_hot_cross_section).parameters["grid"].default) == (g["LOG_PH_E_MIN"], g["LOG_PH_E_MAX"], g["LOG_T_MIN"], g["LOG_T_MAX"])


def test_checkpoint_bytes_follow_the_references_fwrite_order(tmp_path):
    from tests.test_output_cpu import PhotonList, _records
    from mcrat_amd import build, engine
    from mcrat_amd.host import build_host
    build.build()
    host = C.CDLL(build_host.build())
    host.mcrat_host_save_checkpoint.restype = C.c_int
    host.mcrat_host_save_checkpoint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(PhotonList), C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int]
    aos = _records(engine, n=40)
    values = {"angle_size": 8, "restart": b"c", "frame": 201, "frame2": 207, "scatt_frame": 333, "time_now": 66.625, "ph_num": len(aos)}
    l = PhotonList(aos.ctypes.data, None, len(aos), 0, len(aos))
    assert host.mcrat_host_save_checkpoint((str(tmp_path) + "/").encode(), values["frame"], values["frame2"], values["scatt_frame"], values["time_now"],
                                           None, C.byref(l), len(aos), 3000, 5, values["angle_size"], 0) == 0
    raw = (tmp_path / "mc_chkpt_5.dat").read_bytes()
    fmt = {"int": "=i", "char": "=c", "double": "=d"}
    pos = 0
    order = FACTS["checkpoint_fwrite_order"]
    assert [w["variable"] for w in order] == ["angle_size", "restart", "frame", "frame2", "scatt_frame", "time_now", "ph_num", "ph"]
    for w in order:
        if w["sizeof"].startswith("struct photon"):
            assert raw[pos:] == aos.tobytes()                                # one record per slot, in slot order, to the end of the file
            pos = len(raw)
            continue
        (v,) = struct.unpack_from(fmt[w["sizeof"]], raw, pos)
        assert v == values[w["variable"]], (w, v)
        pos += struct.calcsize(fmt[w["sizeof"]])
    assert pos == len(raw)


def test_mcpar_is_read_in_the_references_order(tmp_path):
    from tests.test_host_c import McPar
    from mcrat_amd import build
    from mcrat_amd.host import build_host
    build.build()
    host = C.CDLL(build_host.build())
    host.mcrat_host_read_mcpar.restype = C.c_int
    host.mcrat_host_read_mcpar.argtypes = [C.c_char_p, C.POINTER(McPar)]
    order = [d["dest"] for d in FACTS["mcpar_read_order"]]
    assert order == ["fps", "last_frame", "r0_domain[0]", "r0_domain[1]", "r1_domain[0]", "r1_domain[1]", "r2_domain[0]", "r2_domain[1]", "theta_jmin", "theta_j",
                     "n_theta_j", "frm0[i]", "frm2[i]", "inj_radius[i]", "spect", "min_photons", "max_photons", "restart"]
    # a file whose k-th value (in the reference's read order) is a number that says k: every destination must receive its own
    vals = {"fps": 7.0, "last_frame": 1002, "r0_domain[0]": 3.0, "r0_domain[1]": 4e12, "r1_domain[0]": 5.0, "r1_domain[1]": 6e12, "r2_domain[0]": 7.0,
            "r2_domain[1]": 8e13, "theta_jmin": 0.5, "theta_j": 9.5, "n_theta_j": 2, "frm0[i]": [110, 120], "frm2[i]": [3, 4], "inj_radius[i]": [1.25e11, 2.5e12],
            "spect": "w", "min_photons": 1500, "max_photons": 4500, "restart": "c"}
    groups = [("[Hydro/MHD Simulation Block]", ["fps", "last_frame", ("r0_domain[0]", "r0_domain[1]"), ("r1_domain[0]", "r1_domain[1]"), ("r2_domain[0]", "r2_domain[1]")]),
              ("[MCRaT Injection Angles Block]", ["theta_jmin", "theta_j", "n_theta_j", "frm0[i]", "frm2[i]", "inj_radius[i]"]),
              ("[MCRaT Photon Block]", ["spect", "min_photons", "max_photons"]), ("[Initialization/Continuation Block]", ["restart"])]
    flat = []
    text = ""
    for head, lines in groups:
        text += head + "\n\n"
        for item in lines:
            keys = item if isinstance(item, tuple) else (item,)
            flat += list(keys)
            cell = []
            for k in keys:
                v = vals[k]
                cell.append(" ".join(repr(x) if isinstance(x, float) else str(x) for x in v) if isinstance(v, list) else (("%r" % v) if isinstance(v, float) else str(v)))
            text += " ".join(cell) + "\t# comment\n"
        text += "\n"
    assert flat == order                                                       # the file is laid out in the reference's read order
    (tmp_path / "mc.par").write_text(text)
    p = McPar()
    assert host.mcrat_host_read_mcpar(str(tmp_path / "mc.par").encode(), C.byref(p)) == 0
    assert (p.fps, p.last_frame) == (vals["fps"], vals["last_frame"])
    assert list(p.r0_domain) == [3.0, 4e12] and list(p.r1_domain) == [5.0, 6e12] and list(p.r2_domain) == [7.0, 8e13]
    assert (p.theta_jmin, p.theta_j, p.n_theta_j) == (0.5, 9.5, 2)
    assert [p.frm0[i] for i in range(2)] == [110, 120] and [p.frm2[i] for i in range(2)] == [113, 124]      # start + count, mcrat_io.c:1201
    assert [p.inj_radius[i] for i in range(2)] == [float(np.float32(1.25e11)), float(np.float32(2.5e12))]    # strtof, as the reference (:1211)
    assert (p.spect, p.min_photons, p.max_photons, p.restart) == (b"w", 1500, 4500, b"c")
    host.mcrat_host_free_mcpar(C.byref(p))
