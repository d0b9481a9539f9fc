"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package mcrat_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# struct photon of Src/mcrat.h:142-171 (thermal-only build), 176 bytes
PHOTON_DTYPE = np.dtype(
    [("type", "S1"),
     ("p0", "f8"), ("p1", "f8"), ("p2", "f8"), ("p3", "f8"),
     ("comv_p0", "f8"), ("comv_p1", "f8"), ("comv_p2", "f8"), ("comv_p3", "f8"),
     ("r0", "f8"), ("r1", "f8"), ("r2", "f8"),
     ("s0", "f8"), ("s1", "f8"), ("s2", "f8"), ("s3", "f8"),
     ("num_scatt", "f8"),
     ("recalc_properties", "i4"),
     ("weight", "f8"),
     ("nearest_block_index", "i4"),
     ("time_to_scatter", "f8"),
     ("total_optical_depth", "f8")],
    align=True,
)
assert PHOTON_DTYPE.itemsize == 176

CARTESIAN, SPHERICAL, CYLINDRICAL, POLAR = 0, 1, 2, 3
TWO, TWO_POINT_FIVE, THREE = 0, 1, 2

_dp = C.POINTER(C.c_double)


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("iteration", C.c_uint64), ("stream", C.c_uint32),
                ("ev_state", C.c_uint64), ("n_draws", C.c_uint64),
                ("tape", C.POINTER(C.c_double)), ("tape_n", C.c_int64), ("tape_pos", C.c_int64), ("tape_error", C.c_int)]


class PhotonList(C.Structure):
    _fields_ = [("photons", C.c_void_p), ("sorted_indexes", C.POINTER(C.c_int)),
                ("num_photons", C.c_int), ("num_null_photons", C.c_int), ("list_capacity", C.c_int)]


class Hydro(C.Structure):
    _fields_ = [("num_elements", C.c_int),
                ("r0", _dp), ("r1", _dp), ("r2", _dp),
                ("r0_size", _dp), ("r1_size", _dp), ("r2_size", _dp),
                ("v0", _dp), ("v1", _dp), ("v2", _dp),
                ("dens_lab", _dp), ("temp", _dp), ("gamma", _dp),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2),
                ("fps", C.c_double)]


class Config(C.Structure):
    _fields_ = [("dimensions", C.c_int), ("geometry", C.c_int), ("stokes_switch", C.c_int), ("tau_calculation", C.c_int),
                ("hot_table", C.POINTER(C.c_double)), ("n_ph_e", C.c_int), ("n_t", C.c_int),
                ("log_ph_e_min", C.c_double), ("log_ph_e_max", C.c_double), ("log_t_min", C.c_double), ("log_t_max", C.c_double),
                ("optimised", C.c_int), ("fallback_calls", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_longlong), ("photon_steps", C.c_longlong),
                ("frame_scatt_cnt", C.c_longlong), ("num_photons_find_new_element", C.c_longlong),
                ("not_found", C.c_longlong), ("kn_rejections", C.c_longlong), ("event_draws", C.c_longlong),
                ("last_scattered_index", C.c_int), ("last_time_step", C.c_double),
                ("remaining_time", C.c_double), ("time_now", C.c_double), ("table_fallbacks", C.c_longlong)]


FRAME_FIELDS = ("r0", "r1", "r2", "r0_size", "r1_size", "r2_size", "v0", "v1", "v2", "dens", "dens_lab", "pres", "temp", "gamma", "r", "theta")


class Frame(C.Structure):
    _fields_ = [("num_elements", C.c_int)] + [(f, C.POINTER(C.c_double)) for f in FRAME_FIELDS]


class Slab(C.Structure):
    _fields_ = [("r_inj", C.c_double), ("ph_inj_switch", C.c_int), ("min_r", C.c_double), ("max_r", C.c_double),
                ("min_theta", C.c_double), ("max_theta", C.c_double), ("fps", C.c_double)]


class FlashBlocks(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("coord_stride", C.c_int), ("bsize_stride", C.c_int),
                ("coordinates", C.POINTER(C.c_double)), ("block_size", C.POINTER(C.c_double)), ("node_type", C.POINTER(C.c_int)),
                ("velx", C.POINTER(C.c_double)), ("vely", C.POINTER(C.c_double)), ("dens", C.POINTER(C.c_double)), ("pres", C.POINTER(C.c_double)),
                ("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double), ("cyclosynchrotron", C.c_int)]


class PlutoGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int)] + \
               [(f, C.POINTER(C.c_double)) for f in ("x1", "dx1", "x2", "dx2", "x3", "dx3", "rho", "vx1", "vx2", "vx3", "prs")] + \
               [("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double), ("cyclosynchrotron", C.c_int)]


class ChomboLevel(C.Structure):
    _fields_ = [("n_boxes", C.c_int), ("boxes", C.POINTER(C.c_int)), ("box_offsets", C.POINTER(C.c_int)), ("data_len", C.c_longlong),
                ("prob_domain", C.c_int * 6), ("ref_ratio", C.c_int), ("logr", C.c_int),
                ("dx", C.c_double), ("dombeg1", C.c_double), ("dombeg2", C.c_double), ("dombeg3", C.c_double),
                ("g_x2stretch", C.c_double), ("g_x3stretch", C.c_double)]


class Chombo(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("num_vars", C.c_int), ("levels", C.POINTER(ChomboLevel)), ("var_names", C.POINTER(C.c_char_p)),
                ("data", C.POINTER(C.c_double)), ("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double),
                ("cyclosynchrotron", C.c_int)]


def fill_chombo(raw, level_type, top_type, keep):
    """a ctypes view (oracle's orc_chombo or the engine's mcrat_hip_chombo: same members) of a synth.chombo_raw dict"""
    nl = len(raw["levels"])
    levels = (level_type * nl)()
    for i, lv in enumerate(raw["levels"]):
        boxes = np.ascontiguousarray(lv["boxes"], dtype=np.int32)
        offs = np.ascontiguousarray(lv["box_offsets"], dtype=np.int32)
        keep += [boxes, offs]
        L = levels[i]
        L.n_boxes = boxes.shape[0]
        L.boxes = boxes.ctypes.data_as(C.POINTER(C.c_int))
        L.box_offsets = offs.ctypes.data_as(C.POINTER(C.c_int))
        L.data_len = int(len(lv["data"]))
        for k in range(6):
            L.prob_domain[k] = int(lv["prob_domain"][k]) if k < len(lv["prob_domain"]) else 0
        L.ref_ratio, L.logr = int(lv["ref_ratio"]), int(lv["logr"])
        for k in ("dx", "dombeg1", "dombeg2", "dombeg3", "g_x2stretch", "g_x3stretch"):
            setattr(L, k, float(lv.get(k, 0.0)))
    data = np.ascontiguousarray(np.concatenate([np.asarray(lv["data"], dtype=np.float64) for lv in raw["levels"]]))
    names = (C.c_char_p * len(raw["var_names"]))(*[v.encode() for v in raw["var_names"]])
    keep += [levels, data, names]
    h = top_type()
    h.num_levels, h.num_vars = nl, len(raw["var_names"])
    h.levels = levels
    h.var_names = names
    h.data = data.ctypes.data_as(C.POINTER(C.c_double))
    h.l_scale, h.d_scale, h.p_scale = float(raw.get("l_scale", 1.0)), float(raw.get("d_scale", 1.0)), float(raw.get("p_scale", 1.0))
    return h


class CS(C.Structure):
    """orc_cs: the cyclo-synchrotron switches and the extra hydro columns (oracle_cyclosynch.c)"""
    _fields_ = [("b_field_calc", C.c_int), ("epsilon_b", C.c_double), ("rebin_e_perc", C.c_double), ("dens", C.POINTER(C.c_double)),
                ("B0", C.POINTER(C.c_double)), ("B1", C.POINTER(C.c_double)), ("B2", C.POINTER(C.c_double)),
                ("scatt_frame_number", C.c_int), ("inj_frame_number", C.c_int), ("rebin_ang", C.c_double), ("rebin_ang_phi", C.c_double)]


class CSCounts(C.Structure):
    _fields_ = [("num_cyclosynch_ph_emit", C.c_int), ("scatt_cyclosynch_num_ph", C.c_int), ("frame_abs_cnt", C.c_int), ("rebins", C.c_int),
                ("error", C.c_int), ("n_comptonized", C.c_double), ("pool_weight", C.c_double)]


class Outflow(C.Structure):
    _fields_ = [("simulation_type", C.c_int), ("gamma_infinity", C.c_double), ("lumi", C.c_double), ("r00", C.c_double),
                ("t_comov", C.c_double), ("ddensity", C.c_double), ("theta_j", C.c_double), ("p", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("mcrat_oracle.c", "oracle_ingest.c", "oracle_cyclosynch.c", "mcrat_oracle.h", "oracle_rng.c", "oracle_rng.h")):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        d, i, p = C.c_double, C.c_int, C.c_void_p
        cfgp, hp, lp, rp, sp = C.POINTER(Config), C.POINTER(Hydro), C.POINTER(PhotonList), C.POINTER(Rng), C.POINTER(Stats)
        sig = {
            "orc_philox4x32_10": (None, [C.POINTER(C.c_uint32)] * 3),
            "orc_splitmix64_next": (C.c_uint64, [C.POINTER(C.c_uint64)]),
            "orc_rng_init": (None, [rp, C.c_uint64, C.c_uint32]),
            "orc_rng_set_iteration": (None, [rp, C.c_uint64]),
            "orc_rng_freepath_upos": (d, [rp, C.c_uint32]),
            "orc_rng_freepath_bits": (C.c_uint64, [rp, C.c_uint32]),
            "orc_rng_event_begin": (None, [rp, C.c_uint32]),
            "orc_rng_uniform": (d, [rp]),
            "orc_rng_uniform_pos": (d, [rp]),
            "orc_rng_gaussian": (d, [rp, d]),
            "orc_lorentzBoost": (None, [_dp, _dp, _dp, C.c_char]),
            "orc_zeroNorm": (None, [_dp]),
            "orc_dnrm2": (d, [_dp, i]),
            "orc_mcratCoordinateToHydroCoordinate": (None, [cfgp, _dp, d, d, d]),
            "orc_hydroVectorToCartesian": (None, [cfgp, _dp, d, d, d, d, d, d]),
            "orc_checkInBlock": (i, [cfgp, d, d, d, hp, i]),
            "orc_findContainingBlock": (i, [cfgp, d, d, d, hp]),
            "orc_hydroElementVolume": (d, [cfgp, hp, i]),
            "orc_mullerMatrixRotation": (None, [d, _dp]),
            "orc_findXY": (None, [_dp, _dp, _dp, _dp]),
            "orc_findPhi": (d, [_dp, _dp, _dp, _dp]),
            "orc_stokesRotation": (None, [_dp, _dp, _dp, _dp]),
            "orc_kleinNishinaCrossSection": (d, [d]),
            "orc_bessel_K2": (d, [d]),
            "orc_kleinNishinaScatter": (i, [cfgp, _dp, _dp, d, d, d, rp]),
            "orc_sampleThermalElectron": (d, [d, rp]),
            "orc_sampleElectronTheta": (d, [d, rp]),
            "orc_rotateElectron": (None, [_dp, _dp]),
            "orc_singleThermalElectron": (None, [_dp, d, _dp, rp]),
            "orc_singleScatter": (i, [cfgp, _dp, _dp, _dp, rp]),
            "orc_calculateOpticalDepth": (None, [cfgp, p, hp]),
            "orc_poisson": (C.c_longlong, [rp, d]),
            "orc_rng_stream_begin": (None, [rp, C.c_uint32, C.c_uint32]),
            "orc_photonInjection": (i, [cfgp, C.POINTER(C.c_void_p), C.POINTER(i), _dp, d, d, i, i, C.c_char, d, d, hp, C.c_uint64, C.c_uint32]),
            "orc_free": (None, [C.c_void_p]),
            "orc_getThermalCrossSection": (d, [cfgp, d, d, C.POINTER(i)]),
            "orc_singleMaxwellJuttner": (d, [d, d]),
            "orc_boostedCrossSection": (d, [d, d, d]),
            "orc_calculateTotalThermalCrossSection": (d, [d, d, C.c_longlong, C.c_uint64, i]),
            "orc_createHotCrossSection": (None, [_dp, i, i, d, d, d, d, C.c_longlong, C.c_uint64]),
            "orc_grid_attach": (None, [cfgp, hp]),
            "orc_grid_detach": (None, []),
            "orc_list_init": (None, [lp]), "orc_list_free": (None, [lp]), "orc_list_set": (i, [lp, p, i]), "orc_list_realloc": (i, [lp, i]),
            "orc_list_add": (i, [lp, p, i]), "orc_list_set_null": (i, [lp, i]),
            "orc_calcCyclotronFreq": (d, [d]), "orc_calcEB": (d, [d]), "orc_calcDimlessTheta": (d, [d]), "orc_calcBoundaryE": (d, [d, d]),
            "orc_calcB": (d, [C.POINTER(CS), d, d]), "orc_getMagneticFieldMagnitude": (d, [cfgp, C.POINTER(CS), hp, i]),
            "orc_blackbody_ph_spect": (d, [d, d]), "orc_calcCyclosynchRLimits": (d, [i, i, d, d, i]),
            "orc_qk21": (None, [C.c_void_p, p, d, d, _dp, _dp, _dp, _dp]),
            "orc_qags": (i, [C.c_void_p, p, d, d, d, d, i, _dp, _dp, C.POINTER(i)]),
            "orc_photonEmitCyclosynch": (i, [cfgp, C.POINTER(CS), lp, d, d, i, d, d, hp, rp, i, i, _dp, C.POINTER(i)]),
            "orc_phAbsCyclosynch": (d, [cfgp, C.POINTER(CS), lp, hp, C.POINTER(i), C.POINTER(i)]),
            "orc_saveCheckpoint_convert": (i, [lp]),
            "orc_scatter_frame_cs": (None, [cfgp, C.POINTER(CS), lp, hp, rp, _dp, d, d, d, i, d, d, i, C.c_longlong, sp, C.POINTER(CSCounts)]),
            "orc_rebinCyclosynchCompPhotons": (i, [cfgp, C.POINTER(CS), lp, C.POINTER(i), C.POINTER(i), i]),
            "orc_table_fallbacks": (C.c_longlong, []),
            "orc_reset_table_fallbacks": (None, []),
            "orc_findContainingHydroCell": (i, [cfgp, lp, hp, i, sp]),
            "orc_calcMeanFreePath": (None, [cfgp, lp, hp, rp]),
            "orc_updatePhotonPosition": (None, [lp, d]),
            "orc_photonEvent": (d, [cfgp, lp, d, hp, C.POINTER(i), C.POINTER(C.c_longlong), rp, sp]),
            "orc_averagePhotonEnergy": (d, [lp]),
            "orc_phScattStats": (None, [lp, C.POINTER(i), C.POINTER(i), _dp, _dp]),
            "orc_phMinMax": (None, [lp, _dp, _dp, _dp, _dp]),
            "orc_photon_loop": (None, [cfgp, lp, hp, rp, _dp, _dp, C.POINTER(i), C.c_longlong, C.c_uint64, sp]),
            "orc_flash_select": (i, [cfgp, C.POINTER(FlashBlocks), C.POINTER(Slab), i, C.POINTER(Frame), C.POINTER(i)]),
            "orc_pluto_select": (i, [cfgp, C.POINTER(PlutoGrid), C.POINTER(Slab), i, C.POINTER(Frame), C.POINTER(i)]),
            "orc_chombo_select": (i, [cfgp, C.POINTER(Chombo), C.POINTER(Slab), i, C.POINTER(Frame), C.POINTER(i)]),
            "orc_frame_free": (None, [C.POINTER(Frame)]),
            "orc_outflow_defaults": (None, [i, C.POINTER(Outflow)]),
            "orc_hydro_post_read": (None, [cfgp, C.POINTER(Outflow), C.POINTER(Frame)]),
            "orc_sizeof_photon": (i, []),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        assert L.orc_sizeof_photon() == PHOTON_DTYPE.itemsize
        _lib = L
    return _lib


def vec(a):
    """contiguous float64 array and its double* (keep the array alive)."""
    arr = np.ascontiguousarray(a, dtype=np.float64)
    return arr, arr.ctypes.data_as(_dp)


class OracleHydro:
    """Owns the arrays behind an orc_hydro view.  `frame` is a dict with the
    hydro_dataframe field names (see mcrat_amd.synth)."""
    FIELDS = ("r0", "r1", "r2", "r0_size", "r1_size", "r2_size", "v0", "v1", "v2", "dens_lab", "temp", "gamma")

    def __init__(self, frame):
        self.n = int(frame["num_elements"])
        self.keep = {}
        self.c = Hydro()
        self.c.num_elements = self.n
        for f in self.FIELDS:
            a = frame.get(f)
            if a is None:
                a = np.zeros(self.n)
            arr, ptr = vec(a)
            assert arr.shape == (self.n,), f
            self.keep[f] = arr
            setattr(self.c, f, ptr)
        for k in ("r0_domain", "r1_domain", "r2_domain"):
            dom = frame.get(k, (0.0, 0.0))
            getattr(self.c, k)[0] = float(dom[0])
            getattr(self.c, k)[1] = float(dom[1])
        self.c.fps = float(frame.get("fps", 1.0))


class OraclePhotons:
    """Owns an AoS photon array (PHOTON_DTYPE) and the photonList view on it."""

    def __init__(self, aos):
        self.aos = np.ascontiguousarray(aos, dtype=PHOTON_DTYPE).copy()
        n = len(self.aos)
        self.sorted = np.zeros(n, dtype=np.int32)
        self.c = PhotonList()
        self.c.photons = self.aos.ctypes.data
        self.c.sorted_indexes = self.sorted.ctypes.data_as(C.POINTER(C.c_int))
        nulls = int(np.count_nonzero(self.aos["type"] == b"N"))
        self.c.num_photons = n - nulls
        self.c.num_null_photons = nulls
        self.c.list_capacity = n


def make_config(dimensions, geometry, stokes, hot_table=None, grid=None, optimised=False, fallback_calls=0):
    """hot_table: (N_PH_E + 1, N_T + 1) array of log10(sigma / sigma_T) -> TAU_CALCULATION == TABLE on the grid
    (log_ph_e_min, log_ph_e_max, log_t_min, log_t_max), default the reference's (hot_x_section.h:2-10)."""
    c = Config(int(dimensions), int(geometry), int(bool(stokes)), 1)
    c.optimised = int(bool(optimised))           # same results with an exact cell-search grid (orc_grid_attach) and a prefix sort
    c.fallback_calls = int(fallback_calls)       # samples of a look-up off the table (0: 500 000, hot_x_section.c:348)
    if hot_table is not None:
        import numpy as np
        t = np.ascontiguousarray(hot_table, dtype=np.float64)
        c._keep = t                              # the C side holds a pointer into it
        c.tau_calculation = 2
        c.hot_table = t.ctypes.data_as(C.POINTER(C.c_double))
        c.n_ph_e, c.n_t = t.shape[0] - 1, t.shape[1] - 1
        c.log_ph_e_min, c.log_ph_e_max, c.log_t_min, c.log_t_max = grid if grid is not None else (-12.0, 6.0, -4.0, 4.0)
    return c


def photon_injection(cfg, hydro, r_inj, ph_weight, min_photons, max_photons, spect, theta_min, theta_max, seed, stream=0):
    """orc_photonInjection (mclib.c:9-300) -> (structured array of the injected photons, adjusted weight)"""
    import numpy as np
    L = lib()
    out, n, w = C.c_void_p(), C.c_int(0), C.c_double(0)
    rc = L.orc_photonInjection(C.byref(cfg), C.byref(out), C.byref(n), C.byref(w), float(r_inj), float(ph_weight), int(min_photons),
                               int(max_photons), spect.encode() if isinstance(spect, str) else spect, float(theta_min), float(theta_max),
                               C.byref(hydro.c), int(seed), int(stream))
    if rc != 0:
        raise RuntimeError("orc_photonInjection failed: %d" % rc)
    buf = (C.c_char * (n.value * PHOTON_DTYPE.itemsize)).from_address(out.value)
    a = np.frombuffer(buf, dtype=PHOTON_DTYPE).copy()
    L.orc_free(out)
    return a, w.value


def photon_loop(cfg, photons, hydro, seed, time_now, remaining_time, max_iterations=0,
                iteration_base=0, find_switch=1, stream=0, tape=None, tape_pos=0):
    """Run orc_photon_loop; returns (stats, time_now, remaining, find_switch).
    tape: a float64 array of recorded uniforms in [0,1) -- the random stream as an INPUT (oracle_rng.h, TAPE source), consumed from tape_pos on
    in the reference's call order; the position reached is left in photon_loop.tape_pos, a tape that ran out raises."""
    L = lib()
    rng = Rng()
    if tape is not None:
        import numpy as np
        t = np.ascontiguousarray(tape, dtype=np.float64)
        L.orc_rng_init_tape.argtypes = [C.POINTER(Rng), C.POINTER(C.c_double), C.c_int64]
        L.orc_rng_init_tape(C.byref(rng), t.ctypes.data_as(C.POINTER(C.c_double)), int(t.size))
        rng.tape_pos = int(tape_pos)
    else:
        L.orc_rng_init(C.byref(rng), int(seed), int(stream))
    st = Stats()
    tn, rem, sw = C.c_double(time_now), C.c_double(remaining_time), C.c_int(find_switch)
    L.orc_photon_loop(C.byref(cfg), C.byref(photons.c), C.byref(hydro.c), C.byref(rng),
                      C.byref(tn), C.byref(rem), C.byref(sw), int(max_iterations), int(iteration_base), C.byref(st))
    if tape is not None:
        if rng.tape_error:
            raise RuntimeError("the tape ran out after %d uniforms" % rng.tape_pos)
        photon_loop.tape_pos = int(rng.tape_pos)
    return st, tn.value, rem.value, sw.value


# ---------------------------------------------------------------------------------------------- hydro ingest (oracle_ingest.c)
def _fill(struct, arrays, keep):
    for k, a in arrays.items():
        ctype = C.c_int if k == "node_type" else C.c_double
        arr = np.ascontiguousarray(a, dtype=np.int32 if k == "node_type" else np.float64)
        keep.append(arr)
        setattr(struct, k, arr.ctypes.data_as(C.POINTER(ctype)))


def outflow(simulation_type, **overrides):
    """the reference's hard-coded constants (analytic_outflows.c:5,65,140), optionally overridden"""
    o = Outflow()
    lib().orc_outflow_defaults(int(simulation_type), C.byref(o))
    for k, v in overrides.items():
        setattr(o, k, float(v))
    return o


def hydro_ingest(cfg, raw, slab, outflow_params=None, max_elem_factor=1000):
    """getHydroData (mcrat_io.c:1898-1990) on buffers instead of files: the reader's expansion, slab selection and derived
    columns, fillHydroCoordinateToSpherical and the analytic-outflow overwrite.  `raw` is a dict: kind "flash"
    (coordinates, block_size, node_type, velx, vely, dens, pres) or "pluto" (nx, ny, nz, x1..dx3, rho, vx1, vx2, vx3, prs),
    plus l_scale, d_scale, p_scale.  `slab`: r_inj, ph_inj_switch, min_r, max_r, min_theta, max_theta, fps.
    -> (dict of numpy columns, elem_factor)"""
    L = lib()
    keep = []
    s = Slab(float(slab["r_inj"]), int(slab["ph_inj_switch"]), float(slab["min_r"]), float(slab["max_r"]),
             float(slab["min_theta"]), float(slab["max_theta"]), float(slab["fps"]))
    out, ef = Frame(), C.c_int(0)
    scales = dict(l_scale=float(raw.get("l_scale", 1.0)), d_scale=float(raw.get("d_scale", 1.0)), p_scale=float(raw.get("p_scale", 1.0)))
    if raw["kind"] == "flash":
        coords = np.ascontiguousarray(raw["coordinates"], dtype=np.float64)
        bsize = np.ascontiguousarray(raw["block_size"], dtype=np.float64)
        b = FlashBlocks()
        b.n_blocks, b.coord_stride, b.bsize_stride = coords.shape[0], coords.shape[1], bsize.shape[1]
        _fill(b, dict(coordinates=coords, block_size=bsize, node_type=raw["node_type"], velx=raw["velx"], vely=raw["vely"],
                      dens=raw["dens"], pres=raw["pres"]), keep)
        b.l_scale, b.d_scale, b.p_scale = scales["l_scale"], scales["d_scale"], scales["p_scale"]
        b.cyclosynchrotron = int(raw.get("cyclosynchrotron", 0))
        rc = L.orc_flash_select(C.byref(cfg), C.byref(b), C.byref(s), int(max_elem_factor), C.byref(out), C.byref(ef))
    elif raw["kind"] == "chombo":
        h = fill_chombo(raw, ChomboLevel, Chombo, keep)
        h.cyclosynchrotron = int(raw.get("cyclosynchrotron", 0))
        rc = L.orc_chombo_select(C.byref(cfg), C.byref(h), C.byref(s), int(max_elem_factor), C.byref(out), C.byref(ef))
    else:
        g = PlutoGrid()
        g.nx, g.ny, g.nz = int(raw["nx"]), int(raw["ny"]), int(raw.get("nz", 1))
        zeros = np.zeros(1)
        _fill(g, {k: (raw[k] if raw.get(k) is not None else zeros) for k in ("x1", "dx1", "x2", "dx2", "x3", "dx3", "rho", "vx1", "vx2", "vx3", "prs")}, keep)
        g.l_scale, g.d_scale, g.p_scale = scales["l_scale"], scales["d_scale"], scales["p_scale"]
        g.cyclosynchrotron = int(raw.get("cyclosynchrotron", 0))
        rc = L.orc_pluto_select(C.byref(cfg), C.byref(g), C.byref(s), int(max_elem_factor), C.byref(out), C.byref(ef))
    if rc != 0:
        L.orc_frame_free(C.byref(out))
        raise RuntimeError("no cell selected up to elem_factor %d" % max_elem_factor)
    L.orc_hydro_post_read(C.byref(cfg), C.byref(outflow_params) if outflow_params is not None else None, C.byref(out))
    n = out.num_elements
    cols = {f: np.ctypeslib.as_array(getattr(out, f), shape=(n,)).copy() for f in FRAME_FIELDS}
    cols["num_elements"] = n
    L.orc_frame_free(C.byref(out))
    return cols, ef.value
