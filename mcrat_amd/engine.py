"""ctypes binding of libmcrat_hip.so (include/mcrat_hip.h) -- the product path.

There is no CPU fallback here: if the shared library is missing or no MI355X is
visible, construction raises.  Nothing under oracle/ is imported by this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcrat_hip.so")

ABI_VERSION = 1
CARTESIAN, SPHERICAL, CYLINDRICAL, POLAR = 0, 1, 2, 3
TWO, TWO_POINT_FIVE, THREE = 0, 1, 2
TAU_DIRECT = 1
TAU_TABLE = 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

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

F8_COLUMNS = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2",
              "s0", "s1", "s2", "s3", "num_scatt", "weight", "time_to_scatter", "total_optical_depth")


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("dimensions", C.c_int), ("geometry", C.c_int),
                ("stokes_switch", C.c_int), ("tau_calculation", C.c_int), ("cyclosynchrotron_switch", C.c_int),
                ("device", C.c_int), ("stream", C.c_void_p), ("rng_stream", C.c_uint32),
                ("iterations_per_sync", C.c_int), ("use_graph", C.c_int), ("profile", C.c_int),
                ("virtual_rank_photons", C.c_int)]


class PhotonList(C.Structure):
    _fields_ = [("photons", C.c_void_p), ("sorted_indexes", _ip),
                ("num_photons", C.c_int), ("num_null_photons", C.c_int), ("list_capacity", C.c_int)]


class PhotonSoA(C.Structure):
    _fields_ = [("n", C.c_int), ("type", C.c_char_p),
                ("p0", _dp), ("p1", _dp), ("p2", _dp), ("p3", _dp),
                ("comv_p0", _dp), ("comv_p1", _dp), ("comv_p2", _dp), ("comv_p3", _dp),
                ("r0", _dp), ("r1", _dp), ("r2", _dp),
                ("s0", _dp), ("s1", _dp), ("s2", _dp), ("s3", _dp),
                ("num_scatt", _dp), ("recalc_properties", _ip), ("weight", _dp),
                ("nearest_block_index", _ip), ("time_to_scatter", _dp), ("total_optical_depth", _dp)]


class Hydro(C.Structure):
    _fields_ = [("num_elements", C.c_int),
                ("r0", _dp), ("r1", _dp), ("r2", _dp),
                ("r0_size", _dp), ("r1_size", _dp), ("r2_size", _dp),
                ("v0", _dp), ("v1", _dp), ("v2", _dp),
                ("dens_lab", _dp), ("temp", _dp), ("gamma", _dp),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2),
                ("fps", C.c_double)]


class FrameStats(C.Structure):
    _fields_ = [("iterations", C.c_longlong), ("photon_steps", C.c_longlong), ("frame_scatt_cnt", C.c_longlong),
                ("num_photons_find_new_element", C.c_longlong), ("not_found", C.c_longlong),
                ("kn_rejections", C.c_longlong), ("rescans", C.c_longlong),
                ("last_scattered_index", C.c_int), ("last_scattered_temp", C.c_double),
                ("last_time_step", C.c_double), ("remaining_time", C.c_double), ("time_now", C.c_double),
                ("step_kernel_ms", C.c_double), ("step_kernel_launches", C.c_longlong), ("event_kernel_ms", C.c_double),
                ("table_fallbacks", C.c_longlong), ("slot_steps", C.c_longlong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FramePlan(C.Structure):
    """mcrat_hip_frame_plan: several hydro frames of every list of a pool in one launch (the frame queue)"""
    _fields_ = [("n_frames", C.c_int), ("chain_clock", C.c_int), ("restore_each_frame", C.c_int), ("capture_frames", C.c_int),
                ("open", C.POINTER(C.c_int)), ("seeds", C.POINTER(C.c_uint64)), ("time_now", _dp), ("remaining_time", _dp), ("frame_end", _dp),
                ("hydro", C.POINTER(C.c_void_p))]


class Slab(C.Structure):
    """mcrat_hip_slab: the selecting arguments of getHydroData (mcrat_io.h:26) + fps and the hydro domains"""
    _fields_ = [("r_inj", C.c_double), ("ph_inj_switch", C.c_int), ("min_r", C.c_double), ("max_r", C.c_double),
                ("min_theta", C.c_double), ("max_theta", C.c_double), ("fps", C.c_double),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2)]


class FlashBlocks(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("coord_stride", C.c_int), ("bsize_stride", C.c_int),
                ("coordinates", _dp), ("block_size", _dp), ("node_type", _ip),
                ("velx", _dp), ("vely", _dp), ("dens", _dp), ("pres", _dp),
                ("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double)]


class PlutoGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int)] + \
               [(f, _dp) for f in ("x1", "dx1", "x2", "dx2", "x3", "dx3", "rho", "vx1", "vx2", "vx3", "prs")] + \
               [("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double)]


class ChomboLevel(C.Structure):
    _fields_ = [("n_boxes", C.c_int), ("boxes", _ip), ("box_offsets", _ip), ("data_len", C.c_longlong),
                ("prob_domain", C.c_int * 6), ("ref_ratio", C.c_int), ("logr", C.c_int),
                ("dx", C.c_double), ("dombeg1", C.c_double), ("dombeg2", C.c_double), ("dombeg3", C.c_double),
                ("g_x2stretch", C.c_double), ("g_x3stretch", C.c_double)]


class Chombo(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("num_vars", C.c_int), ("levels", C.POINTER(ChomboLevel)), ("var_names", C.POINTER(C.c_char_p)),
                ("data", _dp), ("l_scale", C.c_double), ("d_scale", C.c_double), ("p_scale", C.c_double)]


class Cyclosynch(C.Structure):
    """mcrat_hip_cyclosynch: B_FIELD_CALC (0 INTERNAL_E, 1 TOTAL_E, 2 SIMULATION), EPSILON_B and the rebinning / frame fields"""
    _fields_ = [("b_field_calc", C.c_int), ("epsilon_b", C.c_double), ("rebin_e_perc", C.c_double), ("rebin_ang", C.c_double),
                ("rebin_ang_phi", C.c_double), ("scatt_frame_number", C.c_int), ("inj_frame_number", C.c_int)]


class CyclosynchCounts(C.Structure):
    """mcrat_hip_cyclosynch_counts: the counters of mcrat.c:735-878 for one scatter frame"""
    _fields_ = [("num_cyclosynch_ph_emit", C.c_int), ("scatt_cyclosynch_num_ph", C.c_int), ("frame_abs_cnt", C.c_int), ("rebins", C.c_int),
                ("integrals_not_converged", C.c_int), ("pad", C.c_int), ("n_comptonized", C.c_double), ("pool_weight", C.c_double)]


class Outflow(C.Structure):
    _fields_ = [("simulation_type", C.c_int), ("gamma_infinity", C.c_double), ("lumi", C.c_double), ("r00", C.c_double),
                ("t_comov", C.c_double), ("ddensity", C.c_double), ("theta_j", C.c_double), ("p", C.c_double)]


class IngestResult(C.Structure):
    _fields_ = [("num_elements", C.c_int), ("elem_factor", C.c_int), ("cells_read", C.c_longlong)]


HYDRO_COLUMNS = ("r0", "r1", "r2", "r0_size", "r1_size", "r2_size", "v0", "v1", "v2", "dens", "dens_lab", "pres", "temp", "gamma", "r", "theta")


class HydroColumns(C.Structure):
    _fields_ = [("num_elements", C.c_int)] + [(f, _dp) for f in HYDRO_COLUMNS]


OUTPUT_COLUMNS = ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3", "num_scatt", "weight")


class OutputColumns(C.Structure):
    _fields_ = [("count", C.c_int)] + [(f, _dp) for f in OUTPUT_COLUMNS] + [("type", C.c_char_p)]


class PoolInjectList(C.Structure):
    """mcrat_hip_pool_inject_list"""
    _fields_ = [("inject", C.c_int), ("spect", C.c_char), ("min_photons", C.c_int), ("max_photons", C.c_int),
                ("r_inj", C.c_double), ("ph_weight", C.c_double), ("theta_min", C.c_double), ("theta_max", C.c_double),
                ("seed", C.c_uint64), ("num_photons", C.c_int), ("ph_weight_adjusted", C.c_double), ("status", C.c_int)]


class RankSummary(C.Structure):
    """mcrat_hip_rank_summary: the per-frame reductions and printPhotons' count of one list of a rank pool"""
    _fields_ = [("min_r", C.c_double), ("max_r", C.c_double), ("min_theta", C.c_double), ("max_theta", C.c_double),
                ("avg_scatt", C.c_double), ("avg_r", C.c_double), ("avg_energy", C.c_double),
                ("max_scatt", C.c_int), ("min_scatt", C.c_int), ("num_output", C.c_int), ("list_capacity", C.c_int)]


class PoolCsList(C.Structure):
    """mcrat_hip_pool_cs_list: one list's arguments of a cyclo-synchrotron scatter frame of a rank pool"""
    _fields_ = [("open", C.c_int), ("emit_pool", C.c_int), ("scatt_frame_number", C.c_int), ("inj_frame_number", C.c_int), ("seed", C.c_uint64),
                ("time_now", C.c_double), ("remaining_time", C.c_double), ("r_inj", C.c_double), ("ph_weight_suggest", C.c_double),
                ("theta_min", C.c_double), ("theta_max", C.c_double)]


SCIENCE, CYLINDRICAL_OUTFLOW, SPHERICAL_OUTFLOW, STRUCTURED_SPHERICAL_OUTFLOW = 0, 1, 2, 3    # SIMULATION_TYPE, mcrat.h:30-33

# every symbol include/mcrat_hip.h declares: (restype, argtypes)
_ctx = C.c_void_p
MODE_EXACT, MODE_FAST = 0, 1
SYMBOLS = {
    "mcrat_hip_init": (C.c_int, [C.POINTER(_ctx), C.POINTER(Config)]),
    "mcrat_hip_destroy": (None, [_ctx]),
    "mcrat_hip_version": (C.c_char_p, []),
    "mcrat_hip_strerror": (C.c_char_p, [C.c_int]),
    "mcrat_hip_last_error": (C.c_char_p, [_ctx]),
    "mcrat_hip_set_hydro": (C.c_int, [_ctx, C.POINTER(Hydro)]),
    "mcrat_hip_outflow_defaults": (None, [C.c_int, C.POINTER(Outflow)]),
    "mcrat_hip_ingest_flash": (C.c_int, [_ctx, C.POINTER(FlashBlocks), C.POINTER(Slab), C.POINTER(Outflow), C.POINTER(IngestResult)]),
    "mcrat_hip_ingest_pluto": (C.c_int, [_ctx, C.POINTER(PlutoGrid), C.POINTER(Slab), C.POINTER(Outflow), C.POINTER(IngestResult)]),
    "mcrat_hip_ingest_chombo": (C.c_int, [_ctx, C.POINTER(Chombo), C.POINTER(Slab), C.POINTER(Outflow), C.POINTER(IngestResult)]),
    "mcrat_hip_set_hydro_extras": (C.c_int, [_ctx, _dp, _dp, _dp, _dp]),
    "mcrat_hip_emit_cyclosynch_pool": (C.c_int, [_ctx, C.POINTER(Cyclosynch), C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_uint64, _ip, _dp, _ip]),
    "mcrat_hip_rebin_cyclosynch": (C.c_int, [_ctx, C.POINTER(Cyclosynch), C.c_int, _ip, _ip, _ip]),
    "mcrat_hip_scatter_frame_cyclosynch": (C.c_int, [_ctx, C.POINTER(Cyclosynch), _dp, C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_int,
                                                     C.c_double, C.c_double, C.c_double, C.c_int, C.c_longlong, C.POINTER(FrameStats),
                                                     C.POINTER(CyclosynchCounts)]),
    "mcrat_hip_num_photon_slots": (C.c_int, [_ctx]),
    "mcrat_hip_absorb_cyclosynch": (C.c_int, [_ctx, C.POINTER(Cyclosynch), _ip, _ip, _dp]),
    "mcrat_hip_get_hydro": (C.c_int, [_ctx, C.POINTER(HydroColumns)]),
    "mcrat_hip_get_output": (C.c_int, [_ctx, C.POINTER(OutputColumns)]),
    "mcrat_hip_convert_comptonized": (C.c_int, [_ctx, _ip]),
    "mcrat_hip_get_photons_range": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "mcrat_hip_inject_photons": (C.c_int, [_ctx, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char, C.c_double, C.c_double, C.c_double,
                                           C.c_uint64, _ip, _dp]),
    "mcrat_hip_set_hot_cross_section": (C.c_int, [_ctx, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "mcrat_hip_table_fallback_calls": (C.c_int, [_ctx, C.c_int]),
    "mcrat_hip_pool_select_frame": (C.c_int, [_ctx, C.c_int]),
    "mcrat_hip_create_hot_cross_section": (C.c_int, [_ctx, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_longlong,
                                                     C.c_uint64]),
    "mcrat_hip_set_photons": (C.c_int, [_ctx, C.POINTER(PhotonList)]),
    "mcrat_hip_register_host": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "mcrat_hip_unregister_host": (C.c_int, [_ctx, C.c_void_p]),
    "mcrat_hip_get_photons": (C.c_int, [_ctx, C.POINTER(PhotonList)]),
    "mcrat_hip_set_photons_soa": (C.c_int, [_ctx, C.POINTER(PhotonSoA)]),
    "mcrat_hip_get_photons_soa": (C.c_int, [_ctx, C.POINTER(PhotonSoA)]),
    "mcrat_hip_num_photon_slots": (C.c_int, [_ctx]),
    "mcrat_hip_pool_inject_photons": (C.c_int, [_ctx, C.c_double, C.POINTER(PoolInjectList)]),
    "mcrat_hip_pool_set_photons": (C.c_int, [_ctx, C.c_int, _ip, C.c_void_p]),
    "mcrat_hip_profile_totals": (C.c_int, [_ctx, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "mcrat_hip_outbox_create": (C.c_int, [_ctx, C.POINTER(C.c_void_p)]),
    "mcrat_hip_outbox_destroy": (None, [C.c_void_p]),
    "mcrat_hip_outbox_post": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "mcrat_hip_outbox_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]),
    "mcrat_hip_bind_thread": (C.c_int, [_ctx]),
    "mcrat_hip_share_hydro": (C.c_int, [_ctx, _ctx]),
    "mcrat_hip_propagate_frame": (C.c_int, [_ctx, _dp, C.c_double, C.c_uint64, C.POINTER(FrameStats)]),
    "mcrat_hip_propagate_frame_mode": (C.c_int, [_ctx, _dp, C.c_double, C.c_uint64, C.c_int, C.c_int, C.POINTER(FrameStats)]),
    "mcrat_hip_fast_cadence": (C.c_int, [_ctx, C.c_int]),
    "mcrat_hip_pool_propagate_frames_fast": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_uint64), _dp, _dp, C.c_int, C.POINTER(FrameStats)]),
    "mcrat_hip_begin_frame": (C.c_int, [_ctx, C.c_uint64, C.c_double, C.c_double]),
    "mcrat_hip_run": (C.c_int, [_ctx, C.c_longlong, C.POINTER(FrameStats)]),
    "mcrat_hip_snapshot_photons": (C.c_int, [_ctx]),
    "mcrat_hip_restore_photons": (C.c_int, [_ctx]),
    "mcrat_hip_num_virtual_ranks": (C.c_int, [_ctx]),
    "mcrat_hip_rank_stats": (C.c_int, [_ctx, C.c_int, C.POINTER(FrameStats)]),
    "mcrat_hip_pool_create": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "mcrat_hip_pool_rank": (C.c_int, [_ctx, C.c_int, C.c_uint32, C.POINTER(_ctx)]),
    "mcrat_hip_pool_summaries": (C.c_int, [_ctx, C.POINTER(RankSummary)]),
    "mcrat_hip_pool_scatter_frames_cyclosynch": (C.c_int, [_ctx, C.POINTER(Cyclosynch), C.c_int, C.c_double, C.POINTER(PoolCsList), C.POINTER(FrameStats),
                                                           C.POINTER(CyclosynchCounts)]),
    "mcrat_hip_pool_begin_frames": (C.c_int, [_ctx, _ip, C.POINTER(C.c_uint64), _dp, _dp]),
    "mcrat_hip_pool_frame_stats": (C.c_int, [_ctx, C.POINTER(FrameStats)]),
    "mcrat_hip_pool_layout": (C.c_int, [_ctx, _ip, _ip]),
    "mcrat_hip_pool_run_frames": (C.c_int, [_ctx, C.POINTER(FramePlan), C.POINTER(FrameStats)]),
    "mcrat_hip_step_locate_sample": (C.c_int, [_ctx, C.c_int]),
    "mcrat_hip_step_event": (C.c_int, [_ctx, C.POINTER(FrameStats)]),
    "mcrat_hip_update_photon_position": (C.c_int, [_ctx, C.c_double]),
    "mcrat_hip_frame_statistics": (C.c_int, [_ctx, C.POINTER(FrameStats)]),
    "mcrat_hip_shared_clock_bytes_per_rank": (C.c_size_t, []),
    "mcrat_hip_shared_clock_attach": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "mcrat_hip_shared_clock_attach_device": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_longlong]),
    "mcrat_hip_shared_clock_peer_buffers": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mcrat_hip_shared_clock_set_peers": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mcrat_hip_shared_clock_exchange_push": (C.c_int, [_ctx]),
    "mcrat_hip_shared_clock_exchange_wait": (C.c_int, [_ctx]),
    "mcrat_hip_set_rng_tape": (C.c_int, [_ctx, _dp, C.c_longlong]),
    "mcrat_hip_rng_tape_position": (C.c_int, [_ctx, C.POINTER(C.c_longlong), _ip]),
    "mcrat_hip_shared_clock_exchange": (C.c_int, [_ctx]),
    "mcrat_hip_shared_clock_reset_exchange": (C.c_int, [_ctx]),
    "mcrat_hip_shared_clock_buffers": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mcrat_hip_shared_clock_propose": (C.c_int, [_ctx]),
    "mcrat_hip_shared_clock_resolve": (C.c_int, [_ctx]),
    "mcrat_hip_shared_clock_poll": (C.c_int, [_ctx, _ip, C.POINTER(FrameStats)]),
    "mcrat_hip_shared_clock_finish": (C.c_int, [_ctx, C.POINTER(FrameStats)]),
    "mcrat_hip_ph_minmax": (C.c_int, [_ctx, _dp, _dp, _dp, _dp]),
    "mcrat_hip_scatt_stats": (C.c_int, [_ctx, _ip, _ip, _dp, _dp]),
    "mcrat_hip_avg_energy": (C.c_int, [_ctx, _dp]),
    "mcrat_hip_synchronize": (C.c_int, [_ctx]),
    "mcrat_hip_device_bytes": (C.c_size_t, [_ctx]),
    "mcrat_hip_eval_function": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp, C.c_uint64]),
    "mcrat_hip_lookup_cell": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp, _ip]),
}

_lib = None


def load_library():
    """dlopen libmcrat_hip.so and bind every symbol; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmcrat_hip.so is missing (%s): build it with `python -m mcrat_amd.build`; "
                "there is no CPU fallback for the photon loop" % LIB_PATH)
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64, and torch finds no GPU if the
        # system copy was mapped first.  Loading torch first makes this library bind to the copy torch uses, so that
        # torch streams / tensors / collectives and the engine share one runtime (shared_clock.py, bench.py).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(os.environ.get("MCRAT_HIP_LIB", LIB_PATH))     # MCRAT_HIP_LIB: another build of the same ABI (A/B timing)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


class McratHipError(RuntimeError):
    pass


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Engine:
    """One context of the HIP photon-loop engine (one per rank / GPU)."""

    def __init__(self, dimensions, geometry, stokes=0, device=0, stream=None, rng_stream=0,
                 iterations_per_sync=0, use_graph=False, profile=False, virtual_rank_photons=0, tau_calculation=TAU_DIRECT,
                 cyclosynchrotron=0):
        self.lib = load_library()
        self.cfg = Config(ABI_VERSION, int(dimensions), int(geometry), int(bool(stokes)), int(tau_calculation), int(cyclosynchrotron),
                          int(device), C.c_void_p(stream) if stream else None, int(rng_stream),
                          int(iterations_per_sync), int(bool(use_graph)), int(bool(profile)), int(virtual_rank_photons))
        self.ctx = _ctx()
        rc = self.lib.mcrat_hip_init(C.byref(self.ctx), C.byref(self.cfg))
        if rc != 0:
            self.ctx = _ctx()
            raise McratHipError("mcrat_hip_init: %s" % self.lib.mcrat_hip_strerror(rc).decode())
        self.n = 0

    def close(self):
        if getattr(self, "ctx", None):
            if getattr(self, "pool", None) is None:      # a view goes with its pool
                self.lib.mcrat_hip_destroy(self.ctx)
            self.ctx = _ctx()
            for v in getattr(self, "views", {}).values():
                v.ctx = _ctx()

    # ---- rank pool: R independent lists (the reference's MPI ranks) in one context
    def pool_create(self, n_ranks, slots_per_rank):
        self._check(self.lib.mcrat_hip_pool_create(self.ctx, int(n_ranks), int(slots_per_rank)), "pool_create")
        for v in getattr(self, "views", {}).values():
            v.ctx = _ctx()
        self.views = {}
        self.n_pool_ranks = int(n_ranks)
        self.n = int(self.lib.mcrat_hip_num_photon_slots(self.ctx))

    def pool_rank(self, rank, rng_stream=0):
        """the view of list `rank`: an Engine whose photons, clock and loop state are that list's"""
        ctx = _ctx()
        self._check(self.lib.mcrat_hip_pool_rank(self.ctx, int(rank), int(rng_stream), C.byref(ctx)), "pool_rank")
        if not hasattr(self, "views"):
            self.views = {}                                      # the pool was laid out from C (mcrat_host_run_ranks)
        v = self.views.get(rank)
        if v is None or v.ctx.value != ctx.value:
            v = Engine.__new__(Engine)
            v.lib, v.cfg, v.ctx, v.n, v.pool = self.lib, self.cfg, ctx, 0, self
            self.views[rank] = v
        v.n = int(self.lib.mcrat_hip_num_photon_slots(ctx))      # the list may have been set from C (mcrat_host_run_ranks)
        if getattr(self, "num_elements", None) is not None:
            v.num_elements = self.num_elements
        return v

    def pool_set_photons(self, ranks, records):
        """mcrat_hip_pool_set_photons: records[j] (numpy arrays of PHOTON_DTYPE, the reference's struct photon) becomes the list of pool rank
        ranks[j] -- one copy over PCIe and one launch for all of them (the views must exist: pool_rank)"""
        n = len(ranks)
        arrs = [np.ascontiguousarray(a, dtype=PHOTON_DTYPE) for a in records]
        lists = (PhotonList * n)()
        for j, a in enumerate(arrs):
            nulls = int(np.count_nonzero(a["type"] == b"N"))
            lists[j] = PhotonList(a.ctypes.data, None, len(a) - nulls, nulls, len(a))
        rk = np.ascontiguousarray(ranks, dtype=np.int32)
        self._check(self.lib.mcrat_hip_pool_set_photons(self.ctx, n, rk.ctypes.data_as(_ip), C.cast(lists, C.c_void_p)), "pool_set_photons")
        for r in ranks:
            if hasattr(self, "views") and r in self.views:
                self.views[r].n = int(self.lib.mcrat_hip_num_photon_slots(self.views[r].ctx))

    def pool_scatter_frames_cyclosynch(self, lists, max_photons, fps, b_field_calc=1, epsilon_b=0.5, rebin_e_perc=0.1, rebin_ang=0.5, rebin_ang_phi=10.0):
        """lists: one dict per list (None: the list sits the frame out) with seed, time_now, remaining_time, r_inj, ph_weight_suggest, theta_min,
        theta_max, emit_pool, scatt_frame_number, inj_frame_number -> ([FrameStats], [CyclosynchCounts])"""
        R = self.n_pool_ranks
        arr = (PoolCsList * R)()
        for r, d in enumerate(lists):
            if d is None:
                continue
            arr[r] = PoolCsList(1, int(d.get("emit_pool", 1)), int(d.get("scatt_frame_number", 0)), int(d.get("inj_frame_number", 0)), int(d["seed"]),
                                float(d["time_now"]), float(d["remaining_time"]), float(d["r_inj"]), float(d["ph_weight_suggest"]), float(d["theta_min"]),
                                float(d["theta_max"]))
        cs = Cyclosynch(int(b_field_calc), float(epsilon_b), float(rebin_e_perc), float(rebin_ang), float(rebin_ang_phi), 0, 0)
        st, cnt = (FrameStats * R)(), (CyclosynchCounts * R)()
        for r, d in enumerate(lists):
            if d is not None:
                cnt[r].scatt_cyclosynch_num_ph = int(d.get("scatt_cyclosynch_num_ph", 0))
        self._check(self.lib.mcrat_hip_pool_scatter_frames_cyclosynch(self.ctx, C.byref(cs), int(max_photons), float(fps), arr, st, cnt),
                    "pool_scatter_frames_cyclosynch")
        return list(st), list(cnt)

    def pool_inject_photons(self, fps, lists):
        """lists: one dict per list (None: no injection) with r_inj, ph_weight, min_photons, max_photons, spect, theta_min, theta_max, seed
        -> [(num_photons, ph_weight_adjusted) or None]"""
        R = self.n_pool_ranks
        arr = (PoolInjectList * R)()
        for r, q in enumerate(lists):
            if q is None:
                continue
            a = arr[r]
            a.inject, a.spect = 1, q["spect"].encode() if isinstance(q["spect"], str) else q["spect"]
            a.min_photons, a.max_photons = int(q["min_photons"]), int(q["max_photons"])
            a.r_inj, a.ph_weight, a.theta_min, a.theta_max, a.seed = float(q["r_inj"]), float(q["ph_weight"]), float(q["theta_min"]), float(q["theta_max"]), int(q["seed"])
        self._check(self.lib.mcrat_hip_pool_inject_photons(self.ctx, float(fps), arr), "pool_inject_photons")
        out = []
        for r, q in enumerate(lists):
            if q is None:
                out.append(None)
                continue
            v = self.views.get(r) if hasattr(self, "views") else None
            if v is not None:
                v.n = int(arr[r].num_photons)
            out.append((int(arr[r].num_photons), float(arr[r].ph_weight_adjusted)))
        return out

    def pool_summaries(self):
        out = (RankSummary * self.n_pool_ranks)()
        self._check(self.lib.mcrat_hip_pool_summaries(self.ctx, out), "pool_summaries")
        return list(out)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise McratHipError("%s: %s (%s)" % (what, self.lib.mcrat_hip_strerror(rc).decode(),
                                                 self.lib.mcrat_hip_last_error(self.ctx).decode()))

    # ---- staging
    def set_hot_cross_section(self, thermal_table, grid=(-12.0, 6.0, -4.0, 4.0)):
        """thermal_table: (N_PH_E + 1, N_T + 1) log10(sigma / sigma_T); grid: LOG_PH_E_MIN/MAX, LOG_T_MIN/MAX (hot_x_section.h:2-10)"""
        t = np.ascontiguousarray(thermal_table, dtype=np.float64)
        self._check(self.lib.mcrat_hip_set_hot_cross_section(self.ctx, t.ctypes.data_as(_dp), t.shape[0] - 1, t.shape[1] - 1,
                                                             float(grid[0]), float(grid[1]), float(grid[2]), float(grid[3])),
                    "set_hot_cross_section")

    def table_fallback_calls(self, calls=0):
        """samples of the integral a look-up off the table takes (hot_x_section.c:348: 500000); calls > 0 sets it"""
        return int(self.lib.mcrat_hip_table_fallback_calls(self.ctx, int(calls)))

    def create_hot_cross_section(self, n_ph_e=220, n_t=80, grid=(-12.0, 6.0, -4.0, 4.0), calls=500000, seed=1):
        """createHotCrossSection (hot_x_section.c:82-133) on the device -> (n_ph_e + 1, n_t + 1) array of log10(sigma / sigma_T)"""
        t = np.empty((n_ph_e + 1, n_t + 1))
        self._check(self.lib.mcrat_hip_create_hot_cross_section(self.ctx, t.ctypes.data_as(_dp), int(n_ph_e), int(n_t), float(grid[0]),
                                                                float(grid[1]), float(grid[2]), float(grid[3]), int(calls), int(seed)),
                    "create_hot_cross_section")
        return t

    def inject_photons(self, r_inj, ph_weight, min_photons, max_photons, spect, theta_min, theta_max, fps, seed):
        """photonInjection (mclib.c:9-300) on the device; returns (number of photons, adjusted weight)"""
        n, w = C.c_int(0), C.c_double(0)
        self._check(self.lib.mcrat_hip_inject_photons(self.ctx, float(r_inj), float(ph_weight), int(min_photons), int(max_photons),
                                                      spect.encode() if isinstance(spect, str) else spect, float(theta_min), float(theta_max),
                                                      float(fps), int(seed), C.byref(n), C.byref(w)), "inject_photons")
        self.n = n.value
        return n.value, w.value

    @staticmethod
    def outflow(simulation_type, **overrides):
        """the constants analytic_outflows.c hard-codes for this SIMULATION_TYPE, optionally overridden"""
        o = Outflow()
        load_library().mcrat_hip_outflow_defaults(int(simulation_type), C.byref(o))
        for k, v in overrides.items():
            setattr(o, k, float(v))
        return o

    @staticmethod
    def _slab(slab):
        s = Slab(float(slab["r_inj"]), int(slab["ph_inj_switch"]), float(slab["min_r"]), float(slab["max_r"]),
                 float(slab["min_theta"]), float(slab["max_theta"]), float(slab["fps"]))
        for k in ("r0_domain", "r1_domain", "r2_domain"):
            dom = slab.get(k, (0.0, 0.0))
            getattr(s, k)[0], getattr(s, k)[1] = float(dom[0]), float(dom[1])
        return s

    def ingest(self, raw, slab, outflow=None):
        """getHydroData (mcrat_io.c:1898-1990) on the device from a reader's buffers: raw["kind"] "flash" (coordinates,
        block_size, node_type, velx, vely, dens, pres) or "pluto" (nx, ny, nz, x1..dx3, rho, vx1, vx2, vx3, prs), with
        l_scale, d_scale, p_scale; slab: r_inj, ph_inj_switch, min_r, max_r, min_theta, max_theta, fps, r*_domain.
        The selected frame is staged for the loop; returns (num_elements, elem_factor, cells_read)."""
        keep, res, s = [], IngestResult(), self._slab(slab)

        def ptr(a, dtype=np.float64, ctype=C.c_double):
            arr = np.ascontiguousarray(a, dtype=dtype)
            keep.append(arr)
            return arr.ctypes.data_as(C.POINTER(ctype))
        op = C.byref(outflow) if outflow is not None else None
        if raw["kind"] == "flash":
            coords, bsize = _f8(raw["coordinates"]), _f8(raw["block_size"])
            b = FlashBlocks(coords.shape[0], coords.shape[1], bsize.shape[1], ptr(coords), ptr(bsize), ptr(raw["node_type"], np.int32, C.c_int),
                            ptr(raw["velx"]), ptr(raw["vely"]), ptr(raw["dens"]), ptr(raw["pres"]),
                            float(raw.get("l_scale", 1.0)), float(raw.get("d_scale", 1.0)), float(raw.get("p_scale", 1.0)))
            self._check(self.lib.mcrat_hip_ingest_flash(self.ctx, C.byref(b), C.byref(s), op, C.byref(res)), "ingest_flash")
        elif raw["kind"] == "chombo":
            nl = len(raw["levels"])
            levels = (ChomboLevel * nl)()
            for i, lv in enumerate(raw["levels"]):
                boxes = np.ascontiguousarray(lv["boxes"], dtype=np.int32)
                L = levels[i]
                L.n_boxes, L.boxes, L.box_offsets = boxes.shape[0], ptr(boxes, np.int32, C.c_int), ptr(lv["box_offsets"], np.int32, C.c_int)
                L.data_len = int(len(lv["data"]))
                for k in range(6):
                    L.prob_domain[k] = int(lv["prob_domain"][k]) if k < len(lv["prob_domain"]) else 0
                L.ref_ratio, L.logr = int(lv["ref_ratio"]), int(lv["logr"])
                for k in ("dx", "dombeg1", "dombeg2", "dombeg3", "g_x2stretch", "g_x3stretch"):
                    setattr(L, k, float(lv.get(k, 0.0)))
            names = (C.c_char_p * len(raw["var_names"]))(*[v.encode() for v in raw["var_names"]])
            keep += [levels, names]
            h = Chombo(nl, len(raw["var_names"]), levels, names, ptr(np.concatenate([_f8(lv["data"]) for lv in raw["levels"]])),
                       float(raw.get("l_scale", 1.0)), float(raw.get("d_scale", 1.0)), float(raw.get("p_scale", 1.0)))
            self._check(self.lib.mcrat_hip_ingest_chombo(self.ctx, C.byref(h), C.byref(s), op, C.byref(res)), "ingest_chombo")
        else:
            g = PlutoGrid()
            g.nx, g.ny, g.nz = int(raw["nx"]), int(raw["ny"]), int(raw.get("nz", 1))
            for k in ("x1", "dx1", "x2", "dx2", "x3", "dx3", "rho", "vx1", "vx2", "vx3", "prs"):
                setattr(g, k, ptr(raw[k]) if raw.get(k) is not None else None)
            g.l_scale, g.d_scale, g.p_scale = float(raw.get("l_scale", 1.0)), float(raw.get("d_scale", 1.0)), float(raw.get("p_scale", 1.0))
            self._check(self.lib.mcrat_hip_ingest_pluto(self.ctx, C.byref(g), C.byref(s), op, C.byref(res)), "ingest_pluto")
        self.num_elements = res.num_elements
        return res.num_elements, res.elem_factor, res.cells_read

    def set_hydro_extras(self, dens=None, B0=None, B1=None, B2=None):
        keep = [None if a is None else _f8(a) for a in (dens, B0, B1, B2)]
        self._check(self.lib.mcrat_hip_set_hydro_extras(self.ctx, *[None if a is None else a.ctypes.data_as(_dp) for a in keep]), "set_hydro_extras")

    def emit_cyclosynch_pool(self, r_inj, ph_weight, maximum_photons, theta_min, theta_max, fps, seed, b_field_calc=1, epsilon_b=0.5,
                             rebin_e_perc=0.1, scatt_frame_number=0, inj_frame_number=0):
        """photonEmitCyclosynch, inject_single_switch = 0 (mc_cyclosynch.c:1200-1460) -> (photons emitted, adjusted weight, integrals not converged)"""
        cs = Cyclosynch(int(b_field_calc), float(epsilon_b), float(rebin_e_perc), 0.5, 10.0, int(scatt_frame_number), int(inj_frame_number))
        n, w, bad = C.c_int(), C.c_double(), C.c_int()
        self._check(self.lib.mcrat_hip_emit_cyclosynch_pool(self.ctx, C.byref(cs), float(r_inj), float(ph_weight), int(maximum_photons), float(theta_min),
                                                            float(theta_max), float(fps), int(seed), C.byref(n), C.byref(w), C.byref(bad)),
                    "emit_cyclosynch_pool")
        self.n = int(self.lib.mcrat_hip_num_photon_slots(self.ctx))          # the list doubles when the pool does not fit
        return n.value, w.value, bad.value

    def scatter_frame_cyclosynch(self, time_now, remaining_time, seed, r_inj, ph_weight_suggest, max_photons, theta_min, theta_max, fps, emit_pool=1,
                                 max_iterations=0, b_field_calc=1, epsilon_b=0.5, rebin_e_perc=0.1, rebin_ang=0.5, rebin_ang_phi=10.0,
                                 scatt_frame_number=0, inj_frame_number=0, scatt_cyclosynch_num_ph=0):
        """main()'s scatter-frame body with CYCLOSYNCHROTRON_SWITCH on (mcrat.c:706-878) -> (time_now, FrameStats, CyclosynchCounts)"""
        cs = Cyclosynch(int(b_field_calc), float(epsilon_b), float(rebin_e_perc), float(rebin_ang), float(rebin_ang_phi), int(scatt_frame_number),
                        int(inj_frame_number))
        st, cnt, tn = FrameStats(), CyclosynchCounts(), C.c_double(time_now)
        cnt.scatt_cyclosynch_num_ph = int(scatt_cyclosynch_num_ph)       # main()'s counter, carried from the previous frame
        self._check(self.lib.mcrat_hip_scatter_frame_cyclosynch(self.ctx, C.byref(cs), C.byref(tn), float(remaining_time), int(seed), float(r_inj),
                                                                float(ph_weight_suggest), int(max_photons), float(theta_min), float(theta_max), float(fps),
                                                                int(emit_pool), int(max_iterations), C.byref(st), C.byref(cnt)),
                    "scatter_frame_cyclosynch")
        self.n = int(self.lib.mcrat_hip_num_photon_slots(self.ctx))
        return tn.value, st, cnt

    def rebin_cyclosynch(self, max_photons, rebin_e_perc=0.1, rebin_ang=0.5, rebin_ang_phi=10.0):
        """rebinCyclosynchCompPhotons (mc_cyclosynch.c:246-712) -> (empty bins, num_cyclosynch_ph_emit, scatt_cyclosynch_num_ph)"""
        cs = Cyclosynch(1, 0.5, float(rebin_e_perc), float(rebin_ang), float(rebin_ang_phi), 0, 0)
        e, a, s = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.mcrat_hip_rebin_cyclosynch(self.ctx, C.byref(cs), int(max_photons), C.byref(e), C.byref(a), C.byref(s)), "rebin_cyclosynch")
        return e.value, a.value, s.value

    def absorb_cyclosynch(self, b_field_calc=1, epsilon_b=0.5):
        """phAbsCyclosynch (mc_cyclosynch.c:1571-1623) -> (num_abs_ph, scatt_cyclosynch_num_ph, absorbed weight)"""
        cs = Cyclosynch(int(b_field_calc), float(epsilon_b), 0.1, 0.5, 10.0, 0, 0)
        a, s, w = C.c_int(), C.c_int(), C.c_double()
        self._check(self.lib.mcrat_hip_absorb_cyclosynch(self.ctx, C.byref(cs), C.byref(a), C.byref(s), C.byref(w)), "absorb_cyclosynch")
        return a.value, s.value, w.value

    def get_hydro(self, num_elements=None):
        """the staged frame's columns (struct hydro_dataframe) as a dict of numpy arrays"""
        n = int(num_elements if num_elements is not None else self.num_elements)
        out, cols = HydroColumns(), {}
        out.num_elements = n
        for f in HYDRO_COLUMNS:
            cols[f] = np.empty(n)
            setattr(out, f, cols[f].ctypes.data_as(_dp))
        self._check(self.lib.mcrat_hip_get_hydro(self.ctx, C.byref(out)), "get_hydro")
        cols = {f: a[:out.num_elements] for f, a in cols.items()}
        cols["num_elements"] = out.num_elements
        return cols

    def set_hydro(self, frame):
        self.num_elements = int(frame["num_elements"])
        n = int(frame["num_elements"])
        keep, h = [], Hydro()
        h.num_elements = n
        for f in ("r0", "r1", "r2", "r0_size", "r1_size", "r2_size", "v0", "v1", "v2", "dens_lab", "temp", "gamma"):
            a = frame.get(f)
            if a is None:
                setattr(h, f, None)
                continue
            a = _f8(a)
            assert a.shape == (n,), f
            keep.append(a)
            setattr(h, f, a.ctypes.data_as(_dp))
        for k in ("r0_domain", "r1_domain", "r2_domain"):
            dom = frame.get(k, (0.0, 0.0))
            getattr(h, k)[0], getattr(h, k)[1] = float(dom[0]), float(dom[1])
        h.fps = float(frame.get("fps", 1.0))
        self._check(self.lib.mcrat_hip_set_hydro(self.ctx, C.byref(h)), "set_hydro")

    def set_photons(self, ph):
        """ph: dict of SoA columns (mcrat_amd.synth layout)."""
        n = int(len(ph["p0"]))
        keep, s = [], PhotonSoA()
        s.n = n
        for f in F8_COLUMNS:
            a = ph.get(f)
            if a is None:
                setattr(s, f, None)
                continue
            a = _f8(a)
            assert a.shape == (n,), f
            keep.append(a)
            setattr(s, f, a.ctypes.data_as(_dp))
        t = np.ascontiguousarray(ph["type"], dtype="S1")
        idx = np.ascontiguousarray(ph["nearest_block_index"], dtype=np.int32)
        rc_ = np.ascontiguousarray(ph["recalc_properties"], dtype=np.int32)
        keep += [t, idx, rc_]
        s.type = t.ctypes.data_as(C.c_char_p)
        s.nearest_block_index = idx.ctypes.data_as(_ip)
        s.recalc_properties = rc_.ctypes.data_as(_ip)
        self._check(self.lib.mcrat_hip_set_photons_soa(self.ctx, C.byref(s)), "set_photons_soa")
        self.n = n

    def get_photons(self):
        n = self.n
        out, s = {}, PhotonSoA()
        s.n = n
        for f in F8_COLUMNS:
            out[f] = np.empty(n, dtype=np.float64)
            setattr(s, f, out[f].ctypes.data_as(_dp))
        out["type"] = np.empty(n, dtype="S1")
        out["nearest_block_index"] = np.empty(n, dtype=np.int32)
        out["recalc_properties"] = np.empty(n, dtype=np.int32)
        s.type = out["type"].ctypes.data_as(C.c_char_p)
        s.nearest_block_index = out["nearest_block_index"].ctypes.data_as(_ip)
        s.recalc_properties = out["recalc_properties"].ctypes.data_as(_ip)
        self._check(self.lib.mcrat_hip_get_photons_soa(self.ctx, C.byref(s)), "get_photons_soa")
        return out

    def get_output(self):
        """printPhotons' arrays (mcrat_io.c:137-181): photons with weight != 0 in slot order, compacted on the device"""
        o = OutputColumns()
        self._check(self.lib.mcrat_hip_get_output(self.ctx, C.byref(o)), "get_output (count)")
        m = o.count
        out = {f: np.empty(m) for f in OUTPUT_COLUMNS}
        out["type"] = np.empty(m, dtype="S1")
        for f in OUTPUT_COLUMNS:
            setattr(o, f, out[f].ctypes.data_as(_dp))
        o.type = out["type"].ctypes.data_as(C.c_char_p)
        self._check(self.lib.mcrat_hip_get_output(self.ctx, C.byref(o)), "get_output")
        assert o.count == m
        return out

    def profile_totals(self):
        """(summed ms, launches) of the loop kernel since the context was created (profile=True contexts)"""
        ms, n = C.c_double(), C.c_longlong()
        self._check(self.lib.mcrat_hip_profile_totals(self.ctx, C.byref(ms), C.byref(n)), "profile_totals")
        return ms.value, n.value

    def outbox_create(self):
        """mcrat_hip_outbox_create: a staging area (device + pinned host) for the frame's records and output columns"""
        b = C.c_void_p()
        self._check(self.lib.mcrat_hip_outbox_create(self.ctx, C.byref(b)), "outbox_create")
        return b

    def outbox_post(self, box, records=True, output=True):
        """stage what saveCheckpoint / printPhotons read and start its copy to the host; the photons may change when this returns"""
        self._check(self.lib.mcrat_hip_outbox_post(self.ctx, box, int(records), int(output)), "outbox_post")

    def outbox_wait(self, box):
        """-> (records or None, output columns dict or None): copies of what has landed in the outbox's pinned memory"""
        rec, n, o = C.c_void_p(), C.c_int(), OutputColumns()
        self._check(self.lib.mcrat_hip_outbox_wait(box, C.byref(rec), C.byref(n), C.byref(o)), "outbox_wait")
        records = None
        if rec.value:
            # (the raw bytes first: numpy copies a structured array member by member and leaves the bytes between the members undefined)
            records = np.frombuffer(C.string_at(rec, PHOTON_DTYPE.itemsize * n.value), dtype=PHOTON_DTYPE) if n.value else np.zeros(0, dtype=PHOTON_DTYPE)
        m = o.count
        cols = {f: np.ctypeslib.as_array(getattr(o, f), shape=(m,)).copy() if m else np.empty(0) for f in OUTPUT_COLUMNS}
        # (the raw pointer: reading a c_char_p member gives a Python bytes COPY that ends at the first NUL, not the address)
        type_ptr = C.c_void_p.from_buffer(o, OutputColumns.type.offset).value
        cols["type"] = np.frombuffer(C.string_at(type_ptr, m), dtype="S1").copy() if m else np.empty(0, dtype="S1")
        return records, cols

    def outbox_destroy(self, box):
        self.lib.mcrat_hip_outbox_destroy(box)

    def convert_comptonized(self):
        """saveCheckpoint's 'k' -> 'c' conversion (mcrat_io.c:896-900) on the resident list; returns the number converted"""
        n = C.c_int()
        self._check(self.lib.mcrat_hip_convert_comptonized(self.ctx, C.byref(n)), "convert_comptonized")
        return n.value

    def get_photons_range(self, first, count):
        a = np.zeros(count, dtype=PHOTON_DTYPE)
        self._check(self.lib.mcrat_hip_get_photons_range(self.ctx, int(first), int(count), a.ctypes.data), "get_photons_range")
        return a

    def set_photons_aos(self, aos, num_null=None):
        """aos: numpy array of PHOTON_DTYPE (the reference's struct photon records); num_null: photon_list->num_null_photons if known"""
        a = np.ascontiguousarray(aos, dtype=PHOTON_DTYPE)
        nulls = int(np.count_nonzero(a["type"] == b"N")) if num_null is None else int(num_null)
        l = PhotonList(a.ctypes.data, None, len(a) - nulls, nulls, len(a))
        self._check(self.lib.mcrat_hip_set_photons(self.ctx, C.byref(l)), "set_photons")
        self.n = len(a)

    def get_photons_aos(self, out=None):
        a = np.zeros(self.n, dtype=PHOTON_DTYPE) if out is None else out
        l = PhotonList(a.ctypes.data, None, self.n, 0, self.n)
        self._check(self.lib.mcrat_hip_get_photons(self.ctx, C.byref(l)), "get_photons")
        return a

    def register_host(self, array):
        """page-lock a numpy array the caller hands to set_photons_aos / get_photons_aos(out=...) repeatedly"""
        self._check(self.lib.mcrat_hip_register_host(self.ctx, array.ctypes.data, array.nbytes), "register_host")

    def unregister_host(self, array):
        self._check(self.lib.mcrat_hip_unregister_host(self.ctx, array.ctypes.data), "unregister_host")

    # ---- the loop
    def set_rng_tape(self, uniforms):
        """the random stream as an input: a float64 array of uniforms in [0,1) consumed in the reference's call order (None: the keyed source)"""
        if uniforms is None:
            self._check(self.lib.mcrat_hip_set_rng_tape(self.ctx, None, 0), "set_rng_tape")
            return
        u = _f8(uniforms)
        self._check(self.lib.mcrat_hip_set_rng_tape(self.ctx, u.ctypes.data_as(_dp), int(u.size)), "set_rng_tape")

    def rng_tape_position(self):
        pos, out = C.c_longlong(0), C.c_int(0)
        self._check(self.lib.mcrat_hip_rng_tape_position(self.ctx, C.byref(pos), C.byref(out)), "rng_tape_position")
        return pos.value, bool(out.value)

    def begin_frame(self, seed, time_now, remaining_time):
        self._check(self.lib.mcrat_hip_begin_frame(self.ctx, int(seed), float(time_now), float(remaining_time)), "begin_frame")

    def run(self, max_iterations=0):
        st = FrameStats()
        self._check(self.lib.mcrat_hip_run(self.ctx, int(max_iterations), C.byref(st)), "run")
        return st

    def propagate_frame(self, time_now, remaining_time, seed):
        st = FrameStats()
        tn = C.c_double(time_now)
        self._check(self.lib.mcrat_hip_propagate_frame(self.ctx, C.byref(tn), float(remaining_time), int(seed), C.byref(st)),
                    "propagate_frame")
        return tn.value, st

    def bind_thread(self):
        """select this engine's device in the calling host thread (HIP's current device is per thread)"""
        self._check(self.lib.mcrat_hip_bind_thread(self.ctx), "bind_thread")

    def share_hydro(self, owner):
        """read `owner`'s staged frame (one copy for several contexts in the same hydro frame); the owner must keep it while shared"""
        self._check(self.lib.mcrat_hip_share_hydro(self.ctx, owner.ctx), "share_hydro")
        self._hydro_owner = owner          # keeps the owner alive as long as this engine

    def propagate_frame_fast(self, time_now, remaining_time, seed, windows=0):
        """MCRAT_HIP_MODE_FAST: every photon through the frame on its own clock (statistically, not sequence-, equivalent)"""
        st = FrameStats()
        tn = C.c_double(time_now)
        self._check(self.lib.mcrat_hip_propagate_frame_mode(self.ctx, C.byref(tn), float(remaining_time), int(seed), MODE_FAST, int(windows),
                                                            C.byref(st)), "propagate_frame_mode")
        return tn.value, st

    def fast_cadence(self, set_windows=0):
        """the list's learnt FAST refresh cadence (windows per frame); set_windows > 0 sets it (a restarted run hands the value back)"""
        return int(self.lib.mcrat_hip_fast_cadence(self.ctx, int(set_windows)))

    def pool_propagate_frames_fast(self, open_, seeds, time_now, remaining_time, windows=0):
        """FAST mode for the lists of the pool, each with its own seed and frame time -> per-list FrameStats"""
        R = self.n_pool_ranks
        o = (C.c_int * R)(*[int(x) for x in open_])
        sd = (C.c_uint64 * R)(*[int(x) for x in seeds])
        t = (C.c_double * R)(*[float(x) for x in time_now])
        rem = (C.c_double * R)(*[float(x) for x in remaining_time])
        st = (FrameStats * R)()
        self._check(self.lib.mcrat_hip_pool_propagate_frames_fast(self.ctx, o, sd, t, rem, int(windows), st), "pool_propagate_frames_fast")
        return list(st)

    def frame_plan(self, open_, seeds, time_now, remaining_time, frame_end=None, chain_clock=False, restore_each_frame=False, hydro=None, capture=False):
        """a mcrat_hip_frame_plan from [n_frames][n_ranks] arrays -> (plan, stats array, the arrays the plan points into);
        hydro: per frame None (the pool's own staged frame) or an Engine holding that frame's staged hydro frame"""
        R = self.n_pool_ranks
        o = np.ascontiguousarray(open_, dtype=np.int32).reshape(-1, R)
        F = o.shape[0]
        sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(F, R)
        t = np.ascontiguousarray(time_now, dtype=np.float64).reshape(F, R)
        rem = np.ascontiguousarray(remaining_time, dtype=np.float64).reshape(F, R)
        fe = None if frame_end is None else np.ascontiguousarray(frame_end, dtype=np.float64).reshape(F, R)
        plan = FramePlan(F, int(bool(chain_clock)), int(bool(restore_each_frame)), int(bool(capture)), o.ctypes.data_as(C.POINTER(C.c_int)),
                         sd.ctypes.data_as(C.POINTER(C.c_uint64)), t.ctypes.data_as(_dp), rem.ctypes.data_as(_dp),
                         fe.ctypes.data_as(_dp) if fe is not None else None, None)
        hy = None
        if hydro is not None:
            hy = (C.c_void_p * F)(*[(h.ctx if h is not None else None) for h in hydro])
            plan.hydro = C.cast(hy, C.POINTER(C.c_void_p))
        return plan, (FrameStats * (F * R))(), (o, sd, t, rem, fe, hy, hydro)

    def pool_run_plan(self, plan, stats):
        """mcrat_hip_pool_run_frames on a prepared plan; stats (frame_plan's array) receives item f * n_ranks + r"""
        self._check(self.lib.mcrat_hip_pool_run_frames(self.ctx, C.byref(plan), stats), "pool_run_frames")

    def pool_run_frames(self, open_, seeds, time_now, remaining_time, frame_end=None, chain_clock=False, restore_each_frame=False, hydro=None, capture=False):
        """mcrat_hip_pool_run_frames: the arrays are [n_frames][n_ranks]; every open list through its frames in ONE launch (the frame queue:
        a list that is through frame f starts f + 1 while others are still in f) -> FrameStats [n_frames][n_ranks]"""
        plan, st, keep = self.frame_plan(open_, seeds, time_now, remaining_time, frame_end, chain_clock, restore_each_frame, hydro, capture)
        self.pool_run_plan(plan, st)
        R, F = self.n_pool_ranks, plan.n_frames
        return [[st[f * R + r] for r in range(R)] for f in range(F)]

    def pool_select_frame(self, frame):
        """the pool's read calls (pool_summaries, get_photons_range, get_output, outbox) on the lists as frame `frame` of the last plan with
        capture=True left them; -1: the live lists again"""
        self._check(self.lib.mcrat_hip_pool_select_frame(self.ctx, int(frame)), "pool_select_frame")

    def snapshot_photons(self):
        self._check(self.lib.mcrat_hip_snapshot_photons(self.ctx), "snapshot_photons")

    def restore_photons(self):
        self._check(self.lib.mcrat_hip_restore_photons(self.ctx), "restore_photons")

    def num_virtual_ranks(self):
        return int(self.lib.mcrat_hip_num_virtual_ranks(self.ctx))

    def rank_stats(self, rank):
        st = FrameStats()
        self._check(self.lib.mcrat_hip_rank_stats(self.ctx, int(rank), C.byref(st)), "rank_stats")
        return st

    def step_locate_sample(self, find_nearest_block_switch):
        self._check(self.lib.mcrat_hip_step_locate_sample(self.ctx, int(find_nearest_block_switch)), "step_locate_sample")

    def frame_statistics(self):
        st = FrameStats()
        self._check(self.lib.mcrat_hip_frame_statistics(self.ctx, C.byref(st)), "frame_statistics")
        return st

    def update_photon_position(self, t):
        self._check(self.lib.mcrat_hip_update_photon_position(self.ctx, float(t)), "update_photon_position")

    def step_event(self):
        st = FrameStats()
        self._check(self.lib.mcrat_hip_step_event(self.ctx, C.byref(st)), "step_event")
        return st

    # ---- one list over several GPUs, one clock (mcrat_amd/shared_clock.py drives these)
    def shared_clock_bytes_per_rank(self):
        return int(self.lib.mcrat_hip_shared_clock_bytes_per_rank())

    def shared_clock_attach(self, world, rank, slot_base, send_ptr=None, recv_ptr=None):
        self._check(self.lib.mcrat_hip_shared_clock_attach(self.ctx, int(world), int(rank), int(slot_base),
                                                           C.c_void_p(send_ptr) if send_ptr else None,
                                                           C.c_void_p(recv_ptr) if recv_ptr else None), "shared_clock_attach")

    def shared_clock_attach_device(self, world, rank, slot_base):
        """device-initiated exchange: library-owned fine-grained buffers -> (recv address, recv bytes, flags address, flags bytes)"""
        self._check(self.lib.mcrat_hip_shared_clock_attach_device(self.ctx, int(world), int(rank), int(slot_base)), "shared_clock_attach_device")
        r, f, rb, fb = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._check(self.lib.mcrat_hip_shared_clock_peer_buffers(self.ctx, C.byref(r), C.byref(rb), C.byref(f), C.byref(fb)), "shared_clock_peer_buffers")
        return r.value, rb.value, f.value, fb.value

    def shared_clock_set_peers(self, peer_recv, peer_flags):
        n = len(peer_recv)
        a = (C.c_void_p * n)(*[C.c_void_p(x) for x in peer_recv])
        b = (C.c_void_p * n)(*[C.c_void_p(x) for x in peer_flags])
        self._check(self.lib.mcrat_hip_shared_clock_set_peers(self.ctx, a, b), "shared_clock_set_peers")

    def shared_clock_exchange_push(self):
        self._check(self.lib.mcrat_hip_shared_clock_exchange_push(self.ctx), "shared_clock_exchange_push")

    def shared_clock_exchange_wait(self):
        self._check(self.lib.mcrat_hip_shared_clock_exchange_wait(self.ctx), "shared_clock_exchange_wait")

    def shared_clock_reset_exchange(self):
        self._check(self.lib.mcrat_hip_shared_clock_reset_exchange(self.ctx), "shared_clock_reset_exchange")

    def shared_clock_propose(self):
        self._check(self.lib.mcrat_hip_shared_clock_propose(self.ctx), "shared_clock_propose")

    def shared_clock_resolve(self):
        self._check(self.lib.mcrat_hip_shared_clock_resolve(self.ctx), "shared_clock_resolve")

    def shared_clock_poll(self):
        st = FrameStats()
        done = C.c_int(0)
        self._check(self.lib.mcrat_hip_shared_clock_poll(self.ctx, C.byref(done), C.byref(st)), "shared_clock_poll")
        return bool(done.value), st

    def shared_clock_finish(self):
        st = FrameStats()
        self._check(self.lib.mcrat_hip_shared_clock_finish(self.ctx, C.byref(st)), "shared_clock_finish")
        return st

    # ---- reductions
    def ph_minmax(self):
        v = [C.c_double() for _ in range(4)]
        self._check(self.lib.mcrat_hip_ph_minmax(self.ctx, *[C.byref(x) for x in v]), "ph_minmax")
        return tuple(x.value for x in v)          # min_r, max_r, min_theta, max_theta

    def scatt_stats(self):
        mx, mn, avg, ravg = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self._check(self.lib.mcrat_hip_scatt_stats(self.ctx, C.byref(mx), C.byref(mn), C.byref(avg), C.byref(ravg)), "scatt_stats")
        return mx.value, mn.value, avg.value, ravg.value

    def avg_energy(self):
        e = C.c_double()
        self._check(self.lib.mcrat_hip_avg_energy(self.ctx, C.byref(e)), "avg_energy")
        return e.value

    def synchronize(self):
        self._check(self.lib.mcrat_hip_synchronize(self.ctx), "synchronize")

    def device_bytes(self):
        return int(self.lib.mcrat_hip_device_bytes(self.ctx))

    FN = dict(kn_cross_section=(1, 1, 1), lorentz_boost_photon=(2, 7, 4), lorentz_boost_electron=(3, 7, 4), stokes_rotation=(4, 13, 4),
              thermal_electron=(5, 5, 4), thermal_electron_wave=(6, 5, 4), electron_and_scatter=(7, 9, 13))

    def eval_function(self, name, rows, seed=0):
        """one device function of physics.hpp on an array of argument rows (mcrat_hip_eval_function); returns the result rows"""
        fn, wi, wo = self.FN[name]
        a = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, wi)
        out = np.empty((a.shape[0], wo))
        self._check(self.lib.mcrat_hip_eval_function(self.ctx, fn, a.shape[0], a.ctypes.data_as(_dp), out.ctypes.data_as(_dp), int(seed)), "eval_function")
        return out

    def lookup_cell(self, a0, a1, a2=None):
        a0, a1 = _f8(a0), _f8(a1)
        a2p = _f8(a2) if a2 is not None else None
        out = np.empty(len(a0), dtype=np.int32)
        self._check(self.lib.mcrat_hip_lookup_cell(self.ctx, len(a0), a0.ctypes.data_as(_dp), a1.ctypes.data_as(_dp),
                                                   a2p.ctypes.data_as(_dp) if a2p is not None else None,
                                                   out.ctypes.data_as(_ip)), "lookup_cell")
        return out
