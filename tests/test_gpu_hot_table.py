"""createHotCrossSection on the device (include/mcrat_hip.h mcrat_hip_create_hot_cross_section; Src/hot_x_section.c:82-133)
against the oracle's restatement with the same keyed samples, and the full-size table's physical limits."""
import ctypes as C
import time

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def test_small_table_equals_the_oracles_sample_for_sample(hip, oracle):
    e = hip.Engine(synth.TWO, synth.CYLINDRICAL, 0)
    grid = (-6.0, 2.0, -3.0, 1.5)
    for calls in (1000, 20000, 100):                    # 100: fewer samples than substreams
        got = e.create_hot_cross_section(8, 5, grid, calls=calls, seed=77)
        want = np.empty((9, 6))
        oracle.lib().orc_createHotCrossSection(want.ctypes.data_as(C.POINTER(C.c_double)), 8, 5, *grid, calls, 77)
        assert np.isfinite(got).all()
        # the same samples, the same integrand; the order of the additions and libm's last ulp differ -- and just above its
        # seam at eps = 1e-3 the reference's Klein-Nishina formula (mcrat_scattering.c:610-615) cancels terms of 2/eps^2 ~ 1e6
        # down to ~1, so one ulp of log(1 + 2 eps) is 1e-10 of the cross section
        assert np.allclose(got, want, rtol=0, atol=1e-9), np.abs(got - want).max()
    assert not np.array_equal(got, e.create_hot_cross_section(8, 5, grid, calls=100, seed=78))      # the seed matters
    with pytest.raises(hip.McratHipError):
        e.create_hot_cross_section(8, 5, (2.0, -6.0, -3.0, 1.5), calls=10)
    e.close()


def test_reference_size_table_in_a_second_with_the_right_limits(hip, oracle):
    """221 x 81 entries x 500 000 samples (hot_x_section.h:2-10, hot_x_section.c:346): the reference's table"""
    e = hip.Engine(synth.TWO, synth.CYLINDRICAL, 0, tau_calculation=hip.TAU_TABLE)
    e.create_hot_cross_section(4, 4, calls=1000)                       # warm-up: code object load
    t0 = time.perf_counter()
    table = e.create_hot_cross_section()                               # the reference's grid, calls and all
    dt = time.perf_counter() - t0
    assert table.shape == (221, 81) and np.isfinite(table).all()
    assert dt < 20.0, dt
    sig = 10.0 ** table
    L = oracle.lib()
    eps = 10.0 ** np.linspace(-12.0, 6.0, 221)
    # Thomson limit at every temperature; cold electrons reproduce sigma_KN(eps); hotter electrons scatter hard photons less
    assert np.allclose(sig[:60, :], 1.0, atol=2.5e-2) and abs(sig[:60, :].mean() - 1.0) < 1e-3      # eps <= 8e-8; Monte-Carlo error of
    kn = np.array([L.orc_kleinNishinaCrossSection(x) for x in eps])                                   # 5e5 uniform samples per entry
    assert np.allclose(sig[:, 0], kn, rtol=2.5e-2)
    assert (np.diff(sig[150, 40:]) < 2e-3).all() and sig[150, 80] < 0.2 * sig[150, 40]
    # a few entries against the oracle's integral of the same samples
    for i, j in ((0, 0), (110, 40), (220, 80), (147, 63)):
        want = L.orc_calculateTotalThermalCrossSection(eps[i], 10.0 ** (-4.0 + 0.1 * j), 500000, 1, i * 81 + j)
        assert sig[i, j] == pytest.approx(want, rel=5e-9)
    # and it drives a TABLE run like any table handed in from the host
    frame, ph, cfg = synth.config2(n_photons=2000, nzc=4, lumi=1e53)
    e.set_hot_cross_section(table)
    e.set_hydro(frame)
    e.set_photons(ph)
    e.begin_frame(4, 0.0, 0.2)
    st = e.run(200)
    assert st.frame_scatt_cnt > 20 and st.table_fallbacks == 0
    e.close()
    print("hot cross-section table, 221 x 81 x 500000 samples: %.3f s" % dt)
