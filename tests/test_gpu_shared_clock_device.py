"""The shared clock with the exchange done by the GPUs themselves (peer writes + a local wait instead of a collective;
mcrat_hip_shared_clock_attach_device / _set_peers / _exchange): several contexts of one process addressing each other's buffers directly,
and two processes that map each other's buffers through hipIpc and run the frame from the host C (mcrat_host_shared_clock_frame with
mcrat_host_exchange_device) -- photons bit-identical to the single list, as with the all-gather."""
import ctypes as C
import os

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


def _shards(ph, world):
    from mcrat_amd import sharding
    return [sharding.shard_photons(ph, world, r, even=True) for r in range(world)]


@pytest.mark.parametrize("world", [1, 2, 4])
def test_contexts_of_one_process_exchange_by_peer_writes(hip, world):
    from mcrat_amd.shared_clock import LocalGroup
    frame, ph, cfg = synth.config2(n_photons=2001, nzc=8, stokes=1, lumi=1e54)
    seed, t0, rem, iters = 0x4D435261, 3.0, 1.0 / frame["fps"], 500
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters)
    grp = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, world), device_exchange=True)
    tn, stats = grp.propagate_frame(t0, rem, seed, max_iterations=iters)
    out = grp.get_photons()
    st = stats[0]
    assert st.iterations == st1.iterations == iters and st.frame_scatt_cnt == st1.frame_scatt_cnt > 100 and st.kn_rejections == st1.kn_rejections
    assert st.time_now == st1.time_now and st.last_scattered_index == st1.last_scattered_index
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(np.asarray(out[k]), np.asarray(single[k]), equal_nan=(k in FLOAT_FIELDS)), k
    # a whole frame, polled every 24 rounds (rounds after the frame's end still exchange: every member runs the same number)
    grp2 = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, world), device_exchange=True)
    e2, single2, st2 = _gpu_run(hip, frame, ph, cfg, 7, 0.0, 0.004)
    tn2, stats2 = grp2.propagate_frame(0.0, 0.004, 7, rounds_per_poll=24)
    assert stats2[0].remaining_time == 0.0 and stats2[0].iterations == st2.iterations and stats2[0].frame_scatt_cnt == st2.frame_scatt_cnt
    out2 = grp2.get_photons()
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(np.asarray(out2[k]), np.asarray(single2[k]), equal_nan=(k in FLOAT_FIELDS)), k
    grp.close()
    grp2.close()


def test_a_peer_that_never_arrives_is_an_error_not_a_hang(hip, monkeypatch):
    monkeypatch.setenv("MCRAT_HIP_SC_WAIT_SPINS", "100000")
    frame, ph, cfg = synth.config1(n_photons=400)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    e.set_hydro(frame)
    e.set_photons({k: (v[:200] if isinstance(v, np.ndarray) else v) for k, v in ph.items()})
    recv, rb, flags, fb = e.shared_clock_attach_device(2, 0, 0)
    with pytest.raises(hip.McratHipError):
        e.begin_frame(1, 0.0, 0.1)
        e.shared_clock_propose()
        e.shared_clock_exchange_push()                      # no peers set
    # the second rank's buffers exist (this process allocates stand-ins) but nobody ever pushes into rank 0's: the wait gives up
    import torch
    other_recv = torch.zeros(rb, dtype=torch.uint8, device="cuda")
    other_flags = torch.zeros(fb, dtype=torch.uint8, device="cuda")
    e.shared_clock_set_peers([recv, other_recv.data_ptr()], [flags, other_flags.data_ptr()])
    before = e.get_photons()
    e.begin_frame(1, 0.0, 0.1)
    import time
    t0 = time.perf_counter()
    e.shared_clock_propose()
    e.shared_clock_exchange_push()
    e.shared_clock_exchange_wait()
    e.shared_clock_resolve()
    e.synchronize()
    one_wait = time.perf_counter() - t0
    # the rounds already queued behind the failure (a captured batch would hold 16-32 of them) return at once: together they take no longer
    # than the one wait that gave up, not a budget of spins each
    t0 = time.perf_counter()
    for _ in range(8):
        e.shared_clock_propose()
        e.shared_clock_exchange_push()
        e.shared_clock_exchange_wait()
        e.shared_clock_resolve()
    e.synchronize()
    assert time.perf_counter() - t0 < max(0.5 * one_wait, 0.05), (one_wait, time.perf_counter() - t0)
    with pytest.raises(hip.McratHipError, match="did not arrive"):
        e.shared_clock_poll()
    # nothing was resolved from a stale buffer: the photons are where the frame found them
    after = e.get_photons()
    for k in ("r0", "r1", "r2", "p0", "p1", "p2", "p3", "num_scatt"):       # (the first round's forced re-location pass has run: cells and comoving momenta are its)
        assert np.array_equal(np.asarray(after[k]), np.asarray(before[k]), equal_nan=True), k
    # reset (every rank, then a barrier between the ranks) and the exchange works again: a one-rank world needs nobody else
    e.shared_clock_reset_exchange()
    with pytest.raises(hip.McratHipError):
        e.shared_clock_propose()                             # the frame was closed by the reset
    e.close()


def _worker(rank, world, tmp, q, variant):
    import time
    import torch
    from mcrat_amd import engine, sharding
    from mcrat_amd.host import binding as B
    host, rccl = B.host(), B.host_rccl()
    frame, ph, cfg = synth.config2(n_photons=1200, nzc=8, stokes=1, lumi=1e54)
    lo, hi = sharding.shard_bounds_even(1200, world, rank)
    stream = torch.cuda.Stream()
    eng = engine.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], stream=stream.cuda_stream)
    eng.set_hydro(frame)
    eng.set_photons(sharding.shard_photons(ph, world, rank, even=True))
    recv, rb, flags, fb = eng.shared_clock_attach_device(world, rank, lo)
    # hand the two buffers to the other process as hipIpc handles through files (an MPI program would MPI_Allgather the 64-byte handles)
    hr, hf = C.create_string_buffer(64), C.create_string_buffer(64)
    assert rccl.mcrat_host_ipc_export(C.c_void_p(recv), hr) == 0 and rccl.mcrat_host_ipc_export(C.c_void_p(flags), hf) == 0
    with open(os.path.join(tmp, "h%d.tmp" % rank), "wb") as f:
        f.write(hr.raw + hf.raw)
    os.rename(os.path.join(tmp, "h%d.tmp" % rank), os.path.join(tmp, "h%d" % rank))
    peer_recv, peer_flags = [0] * world, [0] * world
    for r in range(world):
        if r == rank:
            peer_recv[r], peer_flags[r] = recv, flags
            continue
        path = os.path.join(tmp, "h%d" % r)
        t_end = time.time() + 120
        while not os.path.exists(path):
            assert time.time() < t_end
            time.sleep(0.05)
        raw = open(path, "rb").read()
        a, b = C.c_void_p(), C.c_void_p()
        assert rccl.mcrat_host_ipc_import(raw[:64], C.byref(a)) == 0 and rccl.mcrat_host_ipc_import(raw[64:], C.byref(b)) == 0
        peer_recv[r], peer_flags[r] = a.value, b.value
    eng.shared_clock_set_peers(peer_recv, peer_flags)
    t, st = C.c_double(3.0), engine.FrameStats()
    if variant == "loop":
        cb = C.cast(host.mcrat_host_exchange_device, B.ALLGATHER)
        rc = host.mcrat_host_shared_clock_frame(eng.ctx, world, rank, lo, cb, eng.ctx, C.c_void_p(stream.cuda_stream), C.byref(t), 0.004, 99, 16, C.byref(st))
    else:                                                          # the rounds -- propose, push, wait, resolve -- replayed from a hipGraph
        rc = rccl.mcrat_host_shared_clock_frame_graph(eng.ctx, world, rank, lo, None, C.c_void_p(stream.cuda_stream), C.byref(t), 0.004, 99, 16, C.byref(st))
    out = eng.get_photons()
    q.put((rank, rc, st.iterations, st.frame_scatt_cnt, t.value, {k: np.asarray(out[k]) for k in FLOAT_FIELDS + INT_FIELDS}))
    # keep the mappings alive until the peer is done with this process' buffers
    open(os.path.join(tmp, "done%d" % rank), "w").close()
    t_end = time.time() + 120
    while not all(os.path.exists(os.path.join(tmp, "done%d" % r)) for r in range(world)):
        assert time.time() < t_end
        time.sleep(0.05)


@pytest.mark.parametrize("variant", ["loop", "graph"])
def test_two_processes_map_each_others_buffers_through_hipipc(hip, tmp_path, variant):
    from mcrat_amd.host import binding as B
    if B.host_rccl() is None:
        pytest.skip("libmcrat_hip_host_rccl.so (the hipIpc helpers) is not built in this image")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path), q, variant)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    frame, ph, cfg = synth.config2(n_photons=1200, nzc=8, stokes=1, lumi=1e54)
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, 99, 3.0, 0.004)
    for g in got:
        assert g[1] == 0 and (g[2], g[3]) == (st1.iterations, st1.frame_scatt_cnt) and g[4] == st1.time_now
    for k in FLOAT_FIELDS + INT_FIELDS:
        cat = np.concatenate([got[0][5][k], got[1][5][k]])
        assert np.array_equal(cat, np.asarray(single[k]), equal_nan=(k in FLOAT_FIELDS)), k
