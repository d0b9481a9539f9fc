"""BASELINE.json configs[4] composed at test size: a 3-D PLUTO-Chombo AMR frame in spherical coordinates ingested on the device, the magnetic field
taken from the simulation (B_FIELD_CALC == SIMULATION), cyclo-synchrotron emission and absorption, Stokes parameters on -- through the
host-C rank-pool driver (mcrat_host_run_ranks) for two adopted ranks over three hydro frames: injection on the device, a first scatter
frame without pool, then frames with pool emission, replacement of scattered pool photons, absorption, the checkpoint's type
conversion.  Against the oracle doing every rank's steps on its own: the ingest (orc_chombo_select), photonInjection, orc_scatter_frame_cs
with the counter main() carries from frame to frame, saveCheckpoint's conversion."""
import ctypes as C
import os

import numpy as np
import pytest

from mcrat_amd import synth

pytestmark = pytest.mark.gpu
FPS, LAST = 5.0, 2
DOM = dict(r0_domain=(2e11, 3e12), r1_domain=(0.0, 0.8), r2_domain=(0.0, 2 * np.pi))


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _field(cols):
    """a magnetic field that is a function of the selected cells' coordinates (so that device and oracle build the same one)"""
    r, th = np.asarray(cols["r0"]), np.asarray(cols["r1"])
    b = 3e4 * (1e12 / r) * (1.0 + 0.3 * np.sin(3 * th))
    return [np.ascontiguousarray(0.2 * b), np.ascontiguousarray(0.1 * b), np.ascontiguousarray(b)]


def _slab_dict(s):
    return dict(r_inj=s.r_inj, ph_inj_switch=s.ph_inj_switch, min_r=s.min_r, max_r=s.max_r, min_theta=s.min_theta, max_theta=s.max_theta, fps=s.fps,
                r0_domain=tuple(s.r0_domain), r1_domain=tuple(s.r1_domain), r2_domain=tuple(s.r2_domain))


def test_chombo_3d_simulation_field_cyclosynchrotron_stokes_through_the_rank_pool_driver(hip, oracle, tmp_path):
    from mcrat_amd.host import binding as B
    host = B.host()
    raw = synth.chombo_raw(synth.THREE, synth.SPHERICAL, (2e11, 0.0, 0.0), (3e12, 0.8, 2 * np.pi), (32, 16, 16), seed=5, logr=True)
    jet = hip.Engine.outflow(hip.STRUCTURED_SPHERICAL_OUTFLOW, lumi=5e53, theta_j=0.3)
    o_jet = oracle.outflow(3, lumi=5e53, theta_j=0.3)
    R, base_seed, max_photons = 2, 4242, 1500
    ranks = (B.HostRank * R)()
    for r, k in enumerate(ranks):
        d = str(tmp_path / ("rank%d" % r)) + "/"
        os.makedirs(d)
        k.myid, k.angle_id, k.angle_procs = r, r, R
        k.mc_dir = d.encode()
        k.theta_jmin_thread, k.theta_jmax_thread, k.inj_radius, k.ph_weight_suggest = 0.0, 0.25, 1e12, 1e46
        k.framestart, k.frm2, k.rng_seed, k.rng_stream = r, r, base_seed, 40 + r         # rank 1 joins one frame later
    pool = hip.Engine(synth.THREE, synth.SPHERICAL, 1, cyclosynchrotron=1)
    slabs = []

    def reader(user, ctx, frame, slab):
        s = _slab_dict(slab.contents)
        slabs.append((frame, s))
        pool.ingest(raw, s, jet)
        pool.set_hydro_extras(None, *_field(pool.get_hydro()))
        return 0
    cfg = B.PoolConfig()
    cfg.fps, cfg.last_frm = FPS, LAST
    for k, v in DOM.items():
        getattr(cfg, k)[0], getattr(cfg, k)[1] = v
    cfg.spect, cfg.min_photons, cfg.max_photons, cfg.slots_per_rank = b"b", 300, max_photons, 8 * max_photons
    cfg.get_hydro = B.GET_HYDRO(reader)
    cfg.write_checkpoints, cfg.cyclosynchrotron_switch = 1, 1
    cfg.cs = hip.Cyclosynch(2, 0.5, 0.1, 0.5, 45.0, 0, 0)                                # B_FIELD_CALC == SIMULATION
    assert host.mcrat_host_run_ranks(pool.ctx, ranks, R, C.byref(cfg)) == 0, pool.lib.mcrat_hip_last_error(pool.ctx)
    assert [f for f, s in slabs if s["ph_inj_switch"] == 0] == [0, 1, 2]

    ocfg = oracle.make_config(synth.THREE, synth.SPHERICAL, 1)
    L = oracle.lib()
    emitted = absorbed = 0
    for r, k in enumerate(ranks):
        draws = 0
        inj = [s for f, s in slabs if f == k.framestart and s["ph_inj_switch"] == 1][0]
        cols, _ = oracle.hydro_ingest(ocfg, raw, inj, o_jet)
        H = oracle.OracleHydro(dict(cols, **DOM, fps=FPS))
        ph0, w = oracle.photon_injection(ocfg, H, 1e12, 1e46, 300, max_photons, "b", 0.0, 0.25, B.rank_seed(base_seed, draws), stream=k.rng_stream)
        assert len(ph0) == k.num_photons and w == k.ph_weight
        # from here on both sides carry the same list (the device's injection, equal to the oracle's to 1e-11: a difference there would be
        # amplified by the boosts of three frames) -- a context of its own injects what the rank's view injected
        e2 = hip.Engine(synth.THREE, synth.SPHERICAL, 1, rng_stream=k.rng_stream, cyclosynchrotron=1)
        e2.ingest(raw, inj, jet)
        n2, w2 = e2.inject_photons(1e12, 1e46, 300, max_photons, "b", 0.0, 0.25, FPS, B.rank_seed(base_seed, draws))
        inj_dev = e2.get_photons_aos()
        e2.close()
        assert (n2, w2) == (len(ph0), w)
        assert np.allclose(inj_dev["p0"], ph0["p0"], rtol=1e-10) and np.allclose(inj_dev["r0"], ph0["r0"], rtol=1e-10, atol=1e-2)
        ph0 = inj_dev.astype(oracle.PHOTON_DTYPE)
        draws += 1
        l = oracle.PhotonList()
        L.orc_list_init(C.byref(l))
        assert L.orc_list_set(C.byref(l), ph0.ctypes.data, len(ph0)) == 0
        t, carry = C.c_double(k.framestart / FPS), 0
        for F in range(k.framestart, LAST + 1):
            sl = [s for f, s in slabs if f == F and s["ph_inj_switch"] == 0][0]
            cols, _ = oracle.hydro_ingest(ocfg, raw, sl, o_jet)
            H = oracle.OracleHydro(dict(cols, **DOM, fps=FPS))
            Bf = _field(cols)
            dens = np.ascontiguousarray(cols["dens"])
            ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            cs = oracle.CS(2, 0.5, 0.1, ptr(dens), ptr(Bf[0]), ptr(Bf[1]), ptr(Bf[2]), F, k.framestart, 0.5, 45.0)
            rng = oracle.Rng()
            L.orc_rng_init(C.byref(rng), B.rank_seed(base_seed, draws), k.rng_stream)
            draws += 1
            st, cnt = oracle.Stats(), oracle.CSCounts()
            cnt.scatt_cyclosynch_num_ph = carry
            L.orc_scatter_frame_cs(C.byref(ocfg), C.byref(cs), C.byref(l), C.byref(H.c), C.byref(rng), C.byref(t), (F + 1) / FPS - t.value, 1e12, 1e46,
                                   max_photons, 0.0, 0.25, int(F != k.framestart), 0, C.byref(st), C.byref(cnt))
            assert cnt.error == 0
            carry = cnt.scatt_cyclosynch_num_ph
            L.orc_saveCheckpoint_convert(C.byref(l))
            emitted += cnt.num_cyclosynch_ph_emit
            absorbed += cnt.frame_abs_cnt
        buf = (C.c_char * (l.list_capacity * oracle.PHOTON_DTYPE.itemsize)).from_address(l.photons)
        want = np.frombuffer(buf, dtype=oracle.PHOTON_DTYPE).copy()
        L.orc_list_free(C.byref(l))
        v = pool.pool_rank(r, k.rng_stream)
        got = v.get_photons_aos()
        assert len(got) == len(want) and len(want) > k.num_photons                         # the list grew to take the pool
        assert np.array_equal(got["type"], want["type"]) and np.array_equal(got["weight"], want["weight"])
        assert np.array_equal(got["num_scatt"], want["num_scatt"]) and np.array_equal(got["nearest_block_index"], want["nearest_block_index"])
        for f in ("p0", "p1", "p2", "p3", "r0", "r1", "r2", "s0", "s1", "s2", "s3"):
            scale = np.maximum(np.abs(want["p0"]), 1e-300) if f.startswith("p") else (np.maximum(np.abs(want[f]), 1e9) if f.startswith("r") else 1.0)
            err = np.abs(got[f] - want[f]) / scale
            assert np.all(err <= 1e-9), (r, f, float(err.max()))
        assert k.time_now == pytest.approx(t.value, rel=1e-12) and k.scatt_cyclosynch_num_ph == carry
        # the checkpoint the rank left for its last scatter frame ('i' layout): the list after the conversion, 'k' gone
        old = open(k.mc_dir.decode() + "mc_chkpt_%d.dat_old" % k.angle_id, "rb").read()
        rec = np.frombuffer(old[4 + 1 + 4 + 4:], dtype=hip.PHOTON_DTYPE)
        assert old[4:5] == b"i" and np.array_equal(rec["type"], want["type"]) and b"k" not in set(rec["type"][rec["weight"] != 0].tolist())
    assert emitted > 0 and absorbed > 0
    assert sum(k.cyclosynch_emitted_total for k in ranks) == emitted and sum(k.cyclosynch_absorbed_total for k in ranks) == absorbed
    pool.close()
