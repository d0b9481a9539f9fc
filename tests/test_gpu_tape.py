"""The random stream as an INPUT (mcrat_hip_set_rng_tape / oracle_rng.h TAPE source).

MCRaT draws everything from one sequential stream: per pass one gsl_rng_uniform_pos for every located slot in ascending slot order
(mclib.c:646-675), then photonEvent's draws candidate by candidate (electron.c:81,196,217-233; mcrat_scattering.c:519-574).  With a tape the engine
and the oracle consume the SAME caller-supplied uniforms in that order, so a maintainer who records MCRaT's own stream (tools/ref_harness) can hold
the engine's photons against MCRaT's photon for photon.  Here: engine against oracle on tapes made by numpy, with exact zeros in them
(gsl_rng_uniform_pos redraws on 0) -- integers and the tape position exact, doubles to 1e-9."""
import numpy as np
import pytest

from mcrat_amd import synth
from tests.test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _tape(n, seed, zero_every):
    t = np.random.default_rng(seed).random(n)
    if zero_every:
        t[3::zero_every] = 0.0                        # gsl_rng_uniform_pos must skip these, gsl_rng_uniform must take them
        t[10:13] = 0.0                                # ... several in a row, inside the first pass's free-path draws
    return t


CASES = {
    # (zeros no denser than one in a thousand: a zero that lands on the electron's polar-angle draw puts the electron exactly anti-parallel to the
    # photon, where the scattering's azimuth is defined by rounding noise alone -- 6e-8 per draw with MCRaT's 24-bit generator, not a case to compare)
    "cfg1-cartesian": (lambda: synth.config1(n_photons=1500, n0=32, n1=32), 300, 1000),
    "cfg2-cylindrical-stokes": (lambda: synth.config2(n_photons=3000, nzc=8, stokes=1, lumi=1e54), 400, 1000),
    "cfg3-spherical-stokes": (lambda: synth.config3(n_photons=2500, nr=256, nth=128, lumi=1e54), 400, 0),
    "cfg2-hot-maxwell-juttner": (lambda: synth.config2(n_photons=1200, nzc=8, stokes=0, lumi=1e54, r_inj=1e11), 150, 500),
}


@pytest.mark.parametrize("case", list(CASES))
def test_tape_engine_equals_tape_oracle(hip, oracle, case):
    make, passes, zero_every = CASES[case]
    frame, ph, cfg = make()
    n = len(ph["p0"])
    tape = _tape(passes * (n + 4000) + 100000, 11, zero_every)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=64)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.set_rng_tape(tape)
    rem = 1.0 / frame["fps"]
    e.begin_frame(12345, 0.5, rem)                   # (the seed plays no part with a tape)
    st = e.run(passes)
    pos, ran_out = e.rng_tape_position()
    out = e.get_photons()
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    rst, rtn, rrem, _ = oracle.photon_loop(c, P, H, seed=0, time_now=0.5, remaining_time=rem, max_iterations=passes, tape=tape)
    assert not ran_out
    assert (st.iterations, st.frame_scatt_cnt, st.kn_rejections, st.num_photons_find_new_element) == \
           (rst.iterations, rst.frame_scatt_cnt, rst.kn_rejections, rst.num_photons_find_new_element)
    assert st.last_scattered_index == rst.last_scattered_index
    assert pos == oracle.photon_loop.tape_pos       # the same number of uniforms consumed, zeros skipped included
    assert rst.frame_scatt_cnt > passes // 4
    assert st.time_now == pytest.approx(rtn, rel=1e-12)
    _compare(out, P.aos)
    # a second frame reads on where the first stopped (MCRaT's stream is one sequence across frames)
    e.begin_frame(777, 0.5 + rem, rem)
    st2 = e.run(40)
    pos2, _ = e.rng_tape_position()
    rst2, _, _, _ = oracle.photon_loop(c, P, H, seed=0, time_now=0.5 + rem, remaining_time=rem, max_iterations=40, tape=tape, tape_pos=pos)
    assert (st2.iterations, st2.frame_scatt_cnt) == (rst2.iterations, rst2.frame_scatt_cnt) and pos2 == oracle.photon_loop.tape_pos
    _compare(e.get_photons(), P.aos)
    e.close()


def test_tape_results_depend_on_the_tape_not_on_the_seed_and_keyed_mode_returns(hip):
    frame, ph, cfg = synth.config2(n_photons=2000, nzc=8, stokes=0, lumi=1e54)
    rem = 1.0 / frame["fps"]
    runs = {}
    for name, seed, tseed in (("a", 1, 5), ("b", 2, 5), ("c", 1, 6)):
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
        e.set_hydro(frame)
        e.set_photons(ph)
        e.set_rng_tape(_tape(600000, tseed, 0))
        e.begin_frame(seed, 0.0, rem)
        e.run(150)
        runs[name] = e.get_photons()
        if name == "a":                               # back to the keyed source: as if the tape had never been set
            e.set_rng_tape(None)
            e.set_photons(ph)
            e.begin_frame(1, 0.0, rem)
            e.run(150)
            keyed = e.get_photons()
            k = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
            k.set_hydro(frame)
            k.set_photons(ph)
            k.begin_frame(1, 0.0, rem)
            k.run(150)
            assert np.array_equal(keyed["p0"], k.get_photons()["p0"])
            k.close()
        e.close()
    assert np.array_equal(runs["a"]["p0"], runs["b"]["p0"]) and np.array_equal(runs["a"]["r2"], runs["b"]["r2"])
    assert not np.array_equal(runs["a"]["p0"], runs["c"]["p0"])


def test_tape_is_refused_where_it_has_no_meaning_and_a_short_tape_says_so(hip):
    frame, ph, cfg = synth.config2(n_photons=2000, nzc=8, stokes=0, lumi=1e54)
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], virtual_rank_photons=500)
    with pytest.raises(hip.McratHipError, match="one list"):
        e.set_rng_tape(np.full(10, 0.5))
    e.close()
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    with pytest.raises(hip.McratHipError):
        e.set_rng_tape(np.array([0.2, 1.0]))         # gsl_rng_uniform never returns 1
    e.set_hydro(frame)
    e.set_photons(ph)
    e.set_rng_tape(_tape(2500, 9, 0))                # enough for one pass, not for two
    e.begin_frame(1, 0.0, 0.2)
    e.run(3)
    _, ran_out = e.rng_tape_position()
    assert ran_out
    with pytest.raises(hip.McratHipError, match="one list"):
        e.pool_create(2, 1000)
    e.close()


def _locate_event_draws(oracle, frame, ph, cfg, tape):
    """where, on `tape`, the first pass's event takes the electron's polar-angle draw (electron.c:196) and its last draw (the unpolarised azimuth of
    kleinNishinaScatter, mcrat_scattering.c:561, when the first candidate is accepted): the pass's free-path draws come first, one per located slot
    (mclib.c:646-675), then three gsl_ran_gaussian calls (polar method: pairs until one lies in the unit disc, electron.c:233), phi_e, theta_e"""
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=0, time_now=0.0, remaining_time=1.0 / frame["fps"], max_iterations=1, tape=tape)
    end = oracle.photon_loop.tape_pos
    assert rst.frame_scatt_cnt == 1 and rst.kn_rejections == 0
    pos = int(np.count_nonzero(P.aos["nearest_block_index"] != -1))          # (a tape without zeros: one entry per located slot)
    for _ in range(3):
        while True:
            x, y = -1 + 2 * tape[pos], -1 + 2 * tape[pos + 1]
            pos += 2
            if 0 < x * x + y * y <= 1:
                break
    return pos + 1, end - 1, int(rst.last_scattered_index)


@pytest.mark.parametrize("where", ["electron-polar-angle", "klein-nishina-azimuth"])
def test_a_zero_on_an_ill_conditioned_draw_is_bounded_not_avoided(hip, oracle, where):
    """gsl_rng_uniform returns an exact 0 once in 1.7e7 draws of MCRaT's 24-bit ranlxs0 -- several times in a production run.  On the electron's
    polar-angle draw (sampleElectronTheta, Src/electron.c:177-200) u = 0 puts the electron exactly anti-parallel to the photon: cos(theta) = -1 is
    where the reference's own expression has lost its last digits ((1 - sqrt(1 + b^2 + 2b))/b with b -> 1) and where the scattering's azimuth in
    the electron's frame is defined by rounding noise alone, so engine and oracle -- which round differently in the last place (rcp / rsq + Newton
    against IEEE divisions, physics.hpp) -- cannot agree to 1e-9 on THAT photon; north_star's bar is 1e-5.  The case is tested, not avoided: same
    scatterings, rejections, cells, tape position (integers exact); the scattered photon within 1e-5; every other photon within 1e-9.  A zero on the
    Klein-Nishina azimuth draw (phi = 0, mcrat_scattering.c:561) is well conditioned: 1e-9 everywhere."""
    frame, ph, cfg = synth.config2(n_photons=1500, nzc=8, stokes=0, lumi=1e54)
    tape = np.random.default_rng(4).random(400000)
    theta_pos, az_pos, who = _locate_event_draws(oracle, frame, ph, cfg, tape)
    tape = tape.copy()
    tape[theta_pos if where == "electron-polar-angle" else az_pos] = 0.0
    passes = 3
    rem = 1.0 / frame["fps"]
    e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], iterations_per_sync=8)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.set_rng_tape(tape)
    e.begin_frame(1, 0.0, rem)
    st = e.run(passes)
    pos, ran_out = e.rng_tape_position()
    out = e.get_photons()
    H = oracle.OracleHydro(frame)
    c = oracle.make_config(cfg["dimensions"], cfg["geometry"], cfg["stokes"], optimised=True)
    P = oracle.OraclePhotons(synth.photons_to_aos(ph, oracle.PHOTON_DTYPE))
    rst, _, _, _ = oracle.photon_loop(c, P, H, seed=0, time_now=0.0, remaining_time=rem, max_iterations=passes, tape=tape)
    assert not ran_out and pos == oracle.photon_loop.tape_pos
    assert (st.iterations, st.frame_scatt_cnt, st.kn_rejections, st.num_photons_find_new_element, st.last_scattered_index) == \
           (rst.iterations, rst.frame_scatt_cnt, rst.kn_rejections, rst.num_photons_find_new_element, rst.last_scattered_index)
    assert P.aos["num_scatt"][who] >= 1                                       # the photon whose scattering took the zero
    others = np.ones(len(P.aos), dtype=bool)
    others[who] = False
    _compare({k: np.asarray(v)[others] for k, v in out.items()}, P.aos[others])
    _compare({k: np.asarray(v)[~others] for k, v in out.items()}, P.aos[~others], rtol=1e-5 if where == "electron-polar-angle" else 1e-9)
    e.close()
