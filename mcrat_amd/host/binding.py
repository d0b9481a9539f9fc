"""ctypes view of the host-side C (mcrat_hip_host.h): for tests and bench.py.  MCRaT itself links libmcrat_hip_host.so from C."""
import ctypes as C

from mcrat_amd import engine
from mcrat_amd.host import build_host


class McPar(C.Structure):
    """mcrat_host_mcpar"""
    _fields_ = [("fps", C.c_double), ("last_frame", C.c_int),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2),
                ("theta_jmin", C.c_double), ("theta_j", C.c_double), ("n_theta_j", C.c_int),
                ("frm0", C.POINTER(C.c_int)), ("frm2", C.POINTER(C.c_int)), ("inj_radius", C.POINTER(C.c_double)),
                ("spect", C.c_char), ("min_photons", C.c_int), ("max_photons", C.c_int), ("restart", C.c_char)]


class HostRank(C.Structure):
    """mcrat_host_rank: one adopted MPI rank of the reference"""
    _fields_ = [("myid", C.c_int), ("angle_id", C.c_int), ("angle_procs", C.c_int), ("mc_dir", C.c_char * 1024),
                ("theta_jmin_thread", C.c_double), ("theta_jmax_thread", C.c_double), ("inj_radius", C.c_double),
                ("ph_weight_suggest", C.c_double), ("framestart", C.c_int), ("frm2", C.c_int),
                ("rng_seed", C.c_uint64), ("rng_stream", C.c_uint32), ("fPtr", C.c_void_p),
                ("restrt", C.c_char), ("scatt_framestart", C.c_int), ("time_now_start", C.c_double), ("restart_list", C.POINTER(engine.PhotonList)),
                ("fast_cadence_start", C.c_int),
                ("view", C.c_void_p), ("frame", C.c_int), ("scatt_frame", C.c_int), ("time_now", C.c_double),
                ("num_photons", C.c_int), ("ph_weight", C.c_double), ("seeds_drawn", C.c_longlong),
                ("frame_scatt_cnt_total", C.c_longlong), ("fast_cadence", C.c_int), ("scatt_cyclosynch_num_ph", C.c_int), ("first_scatt_frame", C.c_int),
                ("cyclosynch_emitted_total", C.c_longlong), ("cyclosynch_absorbed_total", C.c_longlong), ("state", C.c_int)]


GET_HYDRO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(engine.Slab))


class PoolConfig(C.Structure):
    """mcrat_host_pool_config"""
    _fields_ = [("fps", C.c_double), ("last_frm", C.c_int),
                ("r0_domain", C.c_double * 2), ("r1_domain", C.c_double * 2), ("r2_domain", C.c_double * 2),
                ("spect", C.c_char), ("min_photons", C.c_int), ("max_photons", C.c_int), ("slots_per_rank", C.c_int),
                ("get_hydro", GET_HYDRO), ("user", C.c_void_p), ("write_checkpoints", C.c_int),
                ("print_photons", C.c_void_p), ("comv_switch", C.c_int), ("stokes_switch", C.c_int), ("save_type", C.c_int),
                ("max_frames", C.c_int), ("cyclosynchrotron_switch", C.c_int), ("cs", engine.Cyclosynch), ("mode", C.c_int), ("fast_windows", C.c_int),
                ("sync_output", C.c_int), ("output_threads", C.c_int), ("stage_ctx", C.c_void_p),
                ("hydro_frames_read", C.c_longlong), ("launches", C.c_longlong), ("two_frame_launches", C.c_longlong),
                ("ms_propagate", C.c_double), ("ms_hydro", C.c_double), ("ms_output", C.c_double),
                ("ms_output_writer", C.c_double), ("ms_output_blocked", C.c_double)]


_host = None
_h5 = None


def host():
    """libmcrat_hip_host.so with prototypes"""
    global _host
    if _host is None:
        engine.load_library()
        lib = C.CDLL(build_host.build())
        PL = C.POINTER(engine.PhotonList)
        lib.mcrat_host_read_mcpar.restype = C.c_int
        lib.mcrat_host_read_mcpar.argtypes = [C.c_char_p, C.POINTER(McPar)]
        lib.mcrat_host_free_mcpar.argtypes = [C.POINTER(McPar)]
        lib.mcrat_host_save_checkpoint.restype = C.c_int
        lib.mcrat_host_save_checkpoint.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, PL, C.c_int,
                                                   C.c_int, C.c_int, C.c_int, C.c_int]
        lib.mcrat_host_read_checkpoint.restype = C.c_int
        lib.mcrat_host_read_checkpoint.argtypes = [C.c_char_p, PL, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                   C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        lib.mcrat_host_rank_seed.restype = C.c_uint64
        lib.mcrat_host_rank_seed.argtypes = [C.c_uint64, C.c_longlong]
        lib.mcrat_host_split_ranks.restype = C.c_int
        lib.mcrat_host_split_ranks.argtypes = [C.POINTER(McPar), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_double, C.c_uint64, C.POINTER(HostRank)]
        lib.mcrat_host_shared_clock_frame.restype = C.c_int
        lib.mcrat_host_shared_clock_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.POINTER(C.c_double), C.c_double, C.c_uint64, C.c_int, C.POINTER(engine.FrameStats)]
        lib.mcrat_host_exchange_device.restype = C.c_int
        lib.mcrat_host_exchange_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.mcrat_host_output_floor.restype = C.c_int
        lib.mcrat_host_output_floor.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.mcrat_host_run_ranks.restype = C.c_int
        lib.mcrat_host_run_ranks.argtypes = [C.c_void_p, C.POINTER(HostRank), C.c_int, C.POINTER(PoolConfig)]
        _host = lib
    return _host


def host_h5():
    """libmcrat_hip_host_h5.so (printPhotons' writer and the HDF5 readers), or None where no HDF5 C library is installed"""
    global _h5
    if _h5 is None:
        path = build_host.build_h5()
        if path is None:
            return None
        lib = C.CDLL(path)
        lib.mcrat_host_print_photons.restype = C.c_int
        lib.mcrat_host_print_photons.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.mcrat_host_print_photon_arrays.restype = C.c_int
        lib.mcrat_host_print_photon_arrays.argtypes = [C.POINTER(engine.OutputColumns), C.c_int, C.c_char_p, C.c_int, C.c_void_p]
        lib.mcrat_host_h5_read.restype = C.c_int
        lib.mcrat_host_h5_read.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _h5 = lib
    return _h5


ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_rccl = None


def host_rccl():
    """libmcrat_hip_host_rccl.so (the shared-clock exchange over RCCL, its rounds in a hipGraph), or None without ROCm's RCCL"""
    global _rccl
    if _rccl is None:
        host()
        path = build_host.build_rccl()
        if path is None:
            return None
        lib = C.CDLL(path)
        lib.mcrat_host_shared_clock_frame_graph.restype = C.c_int
        lib.mcrat_host_shared_clock_frame_graph.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                                            C.c_double, C.c_uint64, C.c_int, C.POINTER(engine.FrameStats)]
        lib.mcrat_host_rccl_comm_single.restype = C.c_int
        lib.mcrat_host_rccl_comm_single.argtypes = [C.POINTER(C.c_void_p)]
        lib.mcrat_host_rccl_comm_destroy.argtypes = [C.c_void_p]
        lib.mcrat_host_ipc_export.restype = C.c_int
        lib.mcrat_host_ipc_export.argtypes = [C.c_void_p, C.c_char_p]
        lib.mcrat_host_ipc_import.restype = C.c_int
        lib.mcrat_host_ipc_import.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.mcrat_host_ipc_close.argtypes = [C.c_void_p]
        _rccl = lib
    return _rccl


def rank_seed(rng_seed, k):
    """mcrat_host_rank_seed: the k-th seed drawn from a rank's generator"""
    return int(host().mcrat_host_rank_seed(int(rng_seed), int(k)))
