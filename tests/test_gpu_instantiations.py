"""EVERY rank_loop_kernel instantiation engine.hip can select, against the oracle.

The loop kernel is a template over DIMENSIONS x GEOMETRY x STOKES x TAU_CALCULATION (the reference's compile-time tuple, Src/mcrat.h:262-427) and over
how a list is run: threads per list (128 / 256 / 512), the list's hot columns in LDS or in HBM/L2 (RESIDENT), the fused pass (FUSE: DIRECT optical
depths, not in spherical geometry, 256 or 512 threads), and -- with the cyclo-synchrotron switch -- the hook of mcrat.c:786-808 inside the loop (CSH;
64 / 128 / 256 threads, covered by tests/test_gpu_pool_cyclosynch.py's block sweep).  Several of these builds spill registers (up to 300 B of scratch per
lane, profiles/r04_kernel_resources.txt), and round 3 met one that wrote a wrong Stokes V after an edit elsewhere: a hand-picked sample of instantiations
is not enough.  Here every (physics tuple) gets ONE oracle trajectory per list (three ragged lists, ~40 passes, < 0.2 s of oracle each) and every launch
form the engine's switches can select (MCRAT_HIP_RANK_BLOCK, MCRAT_HIP_RANK_FUSE, MCRAT_HIP_NO_LDS_LISTS) must reproduce it: integers exact, doubles
1e-9.  Through the C ABI."""
import numpy as np
import pytest

from mcrat_amd import synth
from tests.test_gpu_parity import _compare
from tests.test_gpu_pool import _hot_table, _lists

pytestmark = pytest.mark.gpu

LENS = [137, 300, 64]
PASSES = 40


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _case(dims, geom, stokes):
    n = sum(LENS)
    if dims == synth.TWO:
        if geom == synth.CARTESIAN:
            frame, ph, cfg = synth.config1(n_photons=n, n0=32, n1=32, stokes=stokes)
        elif geom == synth.CYLINDRICAL:
            frame, ph, cfg = synth.config2(n_photons=n, nzc=8, stokes=stokes, lumi=1e54)
        else:
            frame, ph, cfg = synth.config3(n_photons=n, nr=256, nth=128, stokes=stokes, lumi=1e54)
    elif dims == synth.TWO_POINT_FIVE:
        if geom == synth.CARTESIAN:
            frame, ph, cfg = synth.config1(n_photons=n, n0=32, n1=32, stokes=stokes)
            frame["dimensions"] = synth.TWO_POINT_FIVE
            synth.add_toroidal_flow(frame)
            cfg = dict(cfg, dimensions=synth.TWO_POINT_FIVE)
        else:
            frame, ph, cfg = synth.config_25d(geom, n_photons=n, stokes=stokes)
    else:
        if geom == synth.CARTESIAN:
            frame, ph, cfg = synth.config_3d_cartesian(n_photons=n)
        else:
            frame, ph, cfg = synth.config_3d(geom, n_photons=n, stokes=stokes)
    cfg = dict(cfg, stokes=int(stokes))
    return frame, ph, cfg


PAIRS = [(synth.TWO, synth.CARTESIAN), (synth.TWO, synth.CYLINDRICAL), (synth.TWO, synth.SPHERICAL),
         (synth.TWO_POINT_FIVE, synth.CARTESIAN), (synth.TWO_POINT_FIVE, synth.CYLINDRICAL), (synth.TWO_POINT_FIVE, synth.SPHERICAL),
         (synth.THREE, synth.CARTESIAN), (synth.THREE, synth.SPHERICAL), (synth.THREE, synth.POLAR)]
NAMES = {synth.TWO: "2d", synth.TWO_POINT_FIVE: "2.5d", synth.THREE: "3d", }
GEOMS = {synth.CARTESIAN: "cartesian", synth.CYLINDRICAL: "cylindrical", synth.SPHERICAL: "spherical", synth.POLAR: "polar"}


def _forms(table, geom):
    """(threads per list, fused pass, columns in LDS) as launch_rank_loop can be asked for them (kernels.hip)"""
    out = []
    for block in (128, 256, 512):
        fuses = (0, 1) if (not table and geom != synth.SPHERICAL and block != 128) else (0,)
        for fuse in fuses:
            for lds in (1, 0):
                out.append((block, fuse, lds))
    return out


@pytest.mark.parametrize("table", [0, 1], ids=["direct", "table"])
@pytest.mark.parametrize("stokes", [0, 1], ids=["stokes-off", "stokes-on"])
@pytest.mark.parametrize("pair", PAIRS, ids=["%s-%s" % (NAMES[d], GEOMS[g]) for d, g in PAIRS])
def test_every_launch_form_of_a_physics_tuple_equals_the_oracle(hip, oracle, monkeypatch, pair, stokes, table):
    dims, geom = pair
    frame, ph, cfg = _case(dims, geom, stokes)
    subs = _lists(ph, LENS)
    R = len(LENS)
    seeds = [4242 + 31 * r for r in range(R)]
    streams = [7, 19, 3]
    t0, rem = 1.5, 1.0 / frame["fps"]
    kw, okw = {}, {}
    if table:
        kw, okw = dict(tau_calculation=hip.TAU_TABLE), dict(hot_table=_hot_table())
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True, **okw)
    want = []
    for r in range(R):
        P = oracle.OraclePhotons(synth.photons_to_aos(subs[r], oracle.PHOTON_DTYPE))
        rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=seeds[r], time_now=t0, remaining_time=rem, max_iterations=PASSES, stream=streams[r])
        want.append((P.aos.copy(), rst, rtn))
    assert sum(w[1].frame_scatt_cnt for w in want) > 0
    for block, fuse, lds in _forms(table, geom):
        monkeypatch.setenv("MCRAT_HIP_RANK_BLOCK", str(block))
        monkeypatch.setenv("MCRAT_HIP_RANK_FUSE", str(fuse))
        if lds:
            monkeypatch.delenv("MCRAT_HIP_NO_LDS_LISTS", raising=False)
        else:
            monkeypatch.setenv("MCRAT_HIP_NO_LDS_LISTS", "1")
        pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], **kw)
        if table:
            pool.set_hot_cross_section(okw["hot_table"])
        pool.set_hydro(frame)
        pool.pool_create(R, 512)
        for r in range(R):
            v = pool.pool_rank(r, streams[r])
            v.set_photons(subs[r])
            v.begin_frame(seeds[r], t0, rem)
        pool.run(PASSES)
        for r in range(R):
            v = pool.pool_rank(r, streams[r])
            st = v.frame_statistics()
            ref, rst, rtn = want[r]
            what = (block, fuse, lds, r)
            assert (st.iterations, st.frame_scatt_cnt, st.kn_rejections, st.num_photons_find_new_element, st.not_found, st.last_scattered_index) == \
                   (rst.iterations, rst.frame_scatt_cnt, rst.kn_rejections, rst.num_photons_find_new_element, rst.not_found, rst.last_scattered_index), what
            assert st.time_now == pytest.approx(rtn, rel=1e-12), what
            try:
                _compare(v.get_photons(), ref)
            except AssertionError as err:
                raise AssertionError("launch form (threads %d, fuse %d, LDS %d), list %d: %s" % (block, fuse, lds, r, err))
        pool.close()
