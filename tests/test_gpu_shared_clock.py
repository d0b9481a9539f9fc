"""One photon list over several GPUs with one clock (mcrat_hip_shared_clock_*, SURVEY.md 8e exact mode): the
photons must end up BIT-identical to a single context holding the whole list, hence within 1e-9 of the oracle.

The device side is exercised with `world` contexts on the one GPU of the test box exchanging by device copies
(shared_clock.LocalGroup) and, once, with two processes exchanging through a gloo group; on a multi-GPU node the
same calls run over RCCL (bench.py --mode shared-clock).
"""
import os
import socket

import numpy as np
import pytest

from mcrat_amd import synth
from .test_gpu_parity import FLOAT_FIELDS, INT_FIELDS, _compare, _gpu_run, _oracle_run, hip  # noqa: F401

pytestmark = pytest.mark.gpu


def _shards(ph, world):
    from mcrat_amd import sharding
    return [sharding.shard_photons(ph, world, r, even=True) for r in range(world)]


def _assert_bitwise(a, b):
    for k in FLOAT_FIELDS + INT_FIELDS + ("type",):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=(k in FLOAT_FIELDS)), k


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_shared_clock_equals_single_list(hip, oracle, world):
    from mcrat_amd.shared_clock import LocalGroup
    frame, ph, cfg = synth.config2(n_photons=2001, nzc=8, stokes=1, lumi=1e54)
    seed, t0, rem, iters = 0x4D435261, 3.0, 1.0 / frame["fps"], 700
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, seed, t0, rem, iters)
    grp = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, world))
    assert grp.n_total == 2001
    tn, stats = grp.propagate_frame(t0, rem, seed, max_iterations=iters)
    out = grp.get_photons()
    st = stats[0]
    assert st.iterations == st1.iterations == iters
    assert st.frame_scatt_cnt == st1.frame_scatt_cnt > 100
    assert st.kn_rejections == st1.kn_rejections
    assert st.last_scattered_index == st1.last_scattered_index          # a global slot
    assert sum(s.num_photons_find_new_element for s in stats) == st1.num_photons_find_new_element
    assert st.time_now == st1.time_now and st.remaining_time == st1.remaining_time
    for s in stats[1:]:                                                  # the replicated loop state stays identical
        assert (s.iterations, s.frame_scatt_cnt, s.time_now, s.remaining_time, s.last_scattered_index, s.last_time_step) == \
               (st.iterations, st.frame_scatt_cnt, st.time_now, st.remaining_time, st.last_scattered_index, st.last_time_step)
    _assert_bitwise(out, single)
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, t0, rem, iters)
    _compare(out, ref)
    grp.close()


def test_shared_clock_kn_rejection_chains_continue_the_iteration(hip, oracle):
    """MeV photons in a hot plasma: candidates are rejected often, so iterations go past what one round of proposals
    covers and continue in midpass rounds (time_to_scatter re-read beyond the cursor, nothing redrawn)."""
    from mcrat_amd.shared_clock import LocalGroup
    frame, ph, cfg = synth.config1(n_photons=600, n0=16, n1=16)
    frame["temp"] = np.full(frame["num_elements"], 4e9)
    for k in ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
        ph[k] = ph[k] * 300.0
    seed, iters = 77, 600
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, 0.0, 0.2, iters)
    grp = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, 3))
    tn, stats = grp.propagate_frame(0.0, 0.2, seed, max_iterations=iters)
    out = grp.get_photons()
    st = stats[0]
    assert rst.kn_rejections > 200
    assert st.iterations == rst.iterations == iters
    assert st.kn_rejections == rst.kn_rejections and st.frame_scatt_cnt == rst.frame_scatt_cnt
    assert st.last_scattered_index == rst.last_scattered_index
    assert st.rescans > 0                                                # midpass rounds happened
    assert st.time_now == pytest.approx(rtn, rel=1e-12)
    _compare(out, ref)
    grp.close()


def test_shared_clock_whole_frame_and_domain_exit(hip, oracle):
    from mcrat_amd.shared_clock import LocalGroup
    frame, ph, cfg = synth.config1(n_photons=500, n0=16, n1=16)
    frame["r1_domain"] = (0.0, 1e12 + 2.0e9)
    seed, t0, rem = 5, 0.0, 0.1
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, t0, rem)
    grp = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, 2))
    tn, stats = grp.propagate_frame(t0, rem, seed, rounds_per_poll=16)
    out = grp.get_photons()
    assert stats[0].remaining_time == 0.0 and rrem == 0.0
    assert stats[0].iterations == rst.iterations and stats[0].frame_scatt_cnt == rst.frame_scatt_cnt
    assert tn == pytest.approx(t0 + rem, rel=1e-13)
    _compare(out, ref)
    grp.close()


def test_shared_clock_tiny_lists_run_out_of_candidates(hip, oracle):
    """fewer photons than proposal slots, one shard with a single photon: the walk can exhaust the list (the loop of
    mclib.c:1128 ends without a scatter)"""
    from mcrat_amd.shared_clock import LocalGroup
    frame, ph, cfg = synth.config1(n_photons=5, n0=16, n1=16)
    frame["temp"] = np.full(frame["num_elements"], 4e9)
    for k in ("p0", "p1", "p2", "p3", "comv_p0", "comv_p1", "comv_p2", "comv_p3"):
        ph[k] = ph[k] * 3000.0
    n = len(ph["p0"])
    seed = 3
    ref, rst, rtn, rrem = _oracle_run(oracle, frame, ph, cfg, seed, 0.0, 0.05, 60)
    grp = LocalGroup(cfg["dimensions"], cfg["geometry"], cfg["stokes"], frame, _shards(ph, 3 if n >= 5 else 2))
    tn, stats = grp.propagate_frame(0.0, 0.05, seed, max_iterations=60)
    out = grp.get_photons()
    assert stats[0].iterations == rst.iterations
    assert stats[0].kn_rejections == rst.kn_rejections and stats[0].frame_scatt_cnt == rst.frame_scatt_cnt
    _compare(out, ref)
    grp.close()


def test_shared_clock_errors(hip):
    frame, ph, cfg = synth.config1(n_photons=100, n0=8, n1=8)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], 0)
    e.set_hydro(frame)
    e.set_photons(ph)
    with pytest.raises(hip.McratHipError):
        e.shared_clock_propose()                      # not attached
    with pytest.raises(hip.McratHipError):
        e.shared_clock_attach(2, 0, 3)                # odd slot_base
    with pytest.raises(hip.McratHipError):
        e.shared_clock_attach(2, 2, 0)                # rank outside the group
    e.shared_clock_attach(1, 0, 0)
    e.begin_frame(1, 0.0, 0.1)
    with pytest.raises(hip.McratHipError):
        e.run(10)                                     # the local-clock loop is refused once attached
    ev = hip.Engine(cfg["dimensions"], cfg["geometry"], 0, virtual_rank_photons=50)
    with pytest.raises(hip.McratHipError):
        ev.shared_clock_attach(1, 0, 0)


# ------------------------------------------------------------------ two processes, gloo group, one GPU
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from mcrat_amd import sharding
    from mcrat_amd.shared_clock import SharedClock, make_engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame, ph, cfg = synth.config2(n_photons=1200, nzc=8, stokes=1, lumi=1e54)
        lo, hi = sharding.shard_bounds_even(1200, world, rank)
        eng = make_engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], device=0)
        eng.set_hydro(frame)
        eng.set_photons(sharding.shard_photons(ph, world, rank, even=True))
        sc = SharedClock(eng, world, rank, lo)
        assert sc.host_staged
        tn, st = sc.propagate_frame(3.0, 1.0 / frame["fps"], 99, max_iterations=250)
        out = eng.get_photons()
        q.put((rank, lo, hi, st.iterations, st.frame_scatt_cnt, {k: np.asarray(out[k]) for k in FLOAT_FIELDS + INT_FIELDS}))
    finally:
        dist.destroy_process_group()


def test_shared_clock_two_processes_over_a_process_group(hip):
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
    frame, ph, cfg = synth.config2(n_photons=1200, nzc=8, stokes=1, lumi=1e54)
    e, single, st1 = _gpu_run(hip, frame, ph, cfg, 99, 3.0, 1.0 / frame["fps"], 250)
    assert got[0][3] == got[1][3] == st1.iterations and got[0][4] == got[1][4] == st1.frame_scatt_cnt
    for k in FLOAT_FIELDS + INT_FIELDS:
        cat = np.concatenate([got[0][5][k], got[1][5][k]])
        assert np.array_equal(cat, np.asarray(single[k]), equal_nan=(k in FLOAT_FIELDS)), k
