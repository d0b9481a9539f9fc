"""world_size-2 test of the multi-GPU host logic on CPU (gloo).  Each rank propagates its photon shard with
its own clock and RNG stream (here through the oracle, the CPU stand-in for the per-rank engine); the union
must equal what one process obtains by running the two shards one after the other, and the per-frame scalar
exchanges (phMinMax extent, counters, max-over-ranks time) must be exact."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_shard(rank, world, seed=11, iters=150):
    sys.path.insert(0, ROOT)
    import ctypes as C
    from mcrat_amd import sharding, synth
    from oracle import oracle_py as O
    frame, ph, cfg = synth.config1(n_photons=801, n0=16, n1=16)       # 801: the shards differ in size
    mine = sharding.shard_photons(ph, world, rank)
    H = O.OracleHydro(frame)
    P = O.OraclePhotons(synth.photons_to_aos(mine, O.PHOTON_DTYPE))
    c = O.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    st, tn, rem, _ = O.photon_loop(c, P, H, seed=seed, time_now=0.0, remaining_time=0.2, max_iterations=iters, stream=rank)
    L = O.lib()
    a, b, c_, d = (C.c_double() for _ in range(4))
    L.orc_phMinMax(C.byref(P.c), C.byref(a), C.byref(b), C.byref(c_), C.byref(d))
    return P.aos, st, (a.value, b.value, c_.value, d.value)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from mcrat_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    aos, st, mm = _run_shard(rank, world)
    dist.barrier()
    ext = sharding.reduce_minmax(*mm)
    tot = sharding.reduce_counters(st.frame_scatt_cnt, st.photon_steps, st.num_photons_find_new_element, 1.0 + rank)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), aos=aos, ext=np.array(ext), tot=np.array(tot),
             scatt=st.frame_scatt_cnt, steps=st.photon_steps)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_independent_shards(tmp_path):
    from mcrat_amd import sharding
    assert [sharding.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharding.shard_bounds(5, 8, 7) == (5, 5)
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    want = [_run_shard(r, world) for r in range(world)]               # the same shards, one process, one after the other
    for r in range(world):
        for k in want[r][0].dtype.names:
            assert np.array_equal(got[r]["aos"][k], want[r][0][k]), (r, k)
    assert len(got[0]["aos"]) + len(got[1]["aos"]) == 801 and len(got[0]["aos"]) == 401
    # the two ranks ran different random streams on different photons
    assert got[0]["scatt"] > 0 and got[1]["scatt"] > 0
    # scalar exchanges: both ranks hold the same, exact, job-wide values
    ext = (min(want[0][2][0], want[1][2][0]), max(want[0][2][1], want[1][2][1]),
           min(want[0][2][2], want[1][2][2]), max(want[0][2][3], want[1][2][3]))
    for r in range(world):
        assert tuple(got[r]["ext"]) == ext
        assert got[r]["tot"][0] == got[0]["scatt"] + got[1]["scatt"]
        assert got[r]["tot"][1] == got[0]["steps"] + got[1]["steps"]
        assert got[r]["tot"][3] == 2.0                                  # slowest rank


def test_even_shard_bounds_for_the_shared_clock_mode():
    """mcrat_amd.sharding.shard_bounds_even: contiguous, covering, every boundary on an even slot (the shared-clock mode
    pairs slots for its free-path random numbers), sizes within one pair of each other"""
    from mcrat_amd import sharding
    for n in (1, 2, 7, 100, 2001, 1_000_000, 10_000_001):
        for world in (1, 2, 3, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = sharding.shard_bounds_even(n, world, r)
                assert lo == prev and (lo % 2 == 0 or lo == hi) and lo <= hi <= n      # (an empty trailing shard may start at an odd n)
                sizes.append(hi - lo)
                prev = hi
            assert prev == n
            assert max(sizes) - min(sizes) <= 3
