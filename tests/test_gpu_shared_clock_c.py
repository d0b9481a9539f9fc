"""The shared-clock frame driven from the host C (mcrat_host_shared_clock_frame, mcrat_hip_host.c; the RCCL exchange and the hipGraph
of rounds, mcrat_hip_host_rccl.c) -- what a C caller (MCRaT's main) would run -- against the single-list engine: bit-identical."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from mcrat_amd import synth
from tests.test_gpu_parity import FLOAT_FIELDS, INT_FIELDS, _gpu_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _case():
    return synth.config2(n_photons=1200, nzc=8, stokes=1, lumi=1e54)


def test_one_rank_frame_through_the_c_loop_and_through_the_graph(hip):
    import torch
    from mcrat_amd.host import binding as B
    host, rccl = B.host(), B.host_rccl()
    frame, ph, cfg = _case()
    rem = 0.004                                                    # a whole (short) frame: a few hundred passes
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, 99, 3.0, rem)
    assert st1.frame_scatt_cnt > 50 and st1.remaining_time == 0.0
    variants = ["loop"] + (["graph", "graph+rccl", "graph+device"] if rccl is not None else [])
    for variant in variants:
        stream = torch.cuda.Stream()
        eng = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream.cuda_stream)
        eng.set_hydro(frame)
        eng.set_photons(ph)
        if variant == "graph+device":                             # the device-initiated exchange (with itself: one rank) inside the captured rounds
            recv, _, flags, _ = eng.shared_clock_attach_device(1, 0, 0)
            eng.shared_clock_set_peers([recv], [flags])
        t, st = C.c_double(3.0), hip.FrameStats()
        if variant == "loop":
            rc = host.mcrat_host_shared_clock_frame(eng.ctx, 1, 0, 0, None, None, C.c_void_p(stream.cuda_stream), C.byref(t), rem, 99, 16, C.byref(st))
        else:
            comm = C.c_void_p()
            if variant == "graph+rccl":
                assert rccl.mcrat_host_rccl_comm_single(C.byref(comm)) == 0       # ncclAllGather (in place, one rank) inside the captured rounds
            rc = rccl.mcrat_host_shared_clock_frame_graph(eng.ctx, 1, 0, 0, comm, C.c_void_p(stream.cuda_stream), C.byref(t), rem, 99, 24, C.byref(st))
            if comm:
                rccl.mcrat_host_rccl_comm_destroy(comm)
        assert rc == 0, (variant, rc)
        assert (st.iterations, st.frame_scatt_cnt, st.kn_rejections) == (st1.iterations, st1.frame_scatt_cnt, st1.kn_rejections), variant
        assert t.value == st1.time_now
        out = eng.get_photons()
        for k in FLOAT_FIELDS + INT_FIELDS:
            assert np.array_equal(np.asarray(out[k]), np.asarray(single[k]), equal_nan=(k in FLOAT_FIELDS)), (variant, k)
        eng.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from mcrat_amd import engine, sharding
    from mcrat_amd.host import binding as B
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        host = B.host()
        frame, ph, cfg = _case()
        lo, hi = sharding.shard_bounds_even(1200, world, rank)
        stream = torch.cuda.Stream()
        eng = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream.cuda_stream)
        eng.set_hydro(frame)
        eng.set_photons(sharding.shard_photons(ph, world, rank, even=True))
        nb = eng.shared_clock_bytes_per_rank()

        def exchange(user, send, recv, nbytes, strm):
            # the rehearsal's all-gather: through host memory and a gloo group (two processes share the one GPU of the box; RCCL wants a
            # device per rank).  A real run passes mcrat_host_allgather_rccl here.
            assert nbytes == nb
            torch.cuda.synchronize()
            mine = torch.empty(nb, dtype=torch.uint8)
            engine.load_library()
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(mine.data_ptr()), C.c_void_p(send), C.c_size_t(nb), 2)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            cat = torch.cat(parts)
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(recv), C.c_void_p(cat.data_ptr()), C.c_size_t(nb * world), 1)
            return 0
        cb = B.ALLGATHER(exchange)
        t, st = C.c_double(3.0), engine.FrameStats()
        rc = host.mcrat_host_shared_clock_frame(eng.ctx, world, rank, lo, cb, None, C.c_void_p(stream.cuda_stream), C.byref(t), 0.004, 99, 1, C.byref(st))
        out = eng.get_photons()
        q.put((rank, rc, st.iterations, st.frame_scatt_cnt, t.value, {k: np.asarray(out[k]) for k in FLOAT_FIELDS + INT_FIELDS}))
    finally:
        dist.destroy_process_group()


def test_two_processes_through_the_c_loop_with_a_callback_exchange(hip):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    frame, ph, cfg = _case()
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, 99, 3.0, 0.004)
    for g in got:
        assert g[1] == 0 and (g[2], g[3]) == (st1.iterations, st1.frame_scatt_cnt) and g[4] == st1.time_now
    for k in FLOAT_FIELDS + INT_FIELDS:
        cat = np.concatenate([got[0][5][k], got[1][5][k]])
        assert np.array_equal(cat, np.asarray(single[k]), equal_nan=(k in FLOAT_FIELDS)), k
