"""The frame queue (mcrat_hip_pool_run_frames): every list of a rank pool through SEVERAL hydro frames in one launch, a list that is
through frame f starting f + 1 while others are still in f -- the reference's ranks are asynchronous processes, each in its own frame
loop (Src/mcrat.c:457-479, :566-934).  What a list sees must not depend on that: its photons, clocks and counters are bit for bit those of
mcrat_hip_pool_begin_frames / begin_frame + mcrat_hip_run one frame at a time (which tests/test_gpu_pool.py holds against the oracle).
Through the C ABI."""
import numpy as np
import pytest

from mcrat_amd import synth
from tests.test_gpu_parity import FLOAT_FIELDS, INT_FIELDS
from tests.test_gpu_pool import _lists

pytestmark = pytest.mark.gpu

STAT_FIELDS = ("iterations", "photon_steps", "frame_scatt_cnt", "num_photons_find_new_element", "not_found", "kn_rejections", "rescans",
               "last_scattered_index", "last_scattered_temp", "last_time_step", "remaining_time", "time_now")


@pytest.fixture(scope="module")
def hip():
    from mcrat_amd import engine
    engine.load_library()
    return engine


def _pool(hip, frame, cfg, subs, streams, window):
    pool = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    pool.set_hydro(frame)
    pool.pool_create(len(subs), window)
    recs = [synth.photons_to_aos(s, hip.PHOTON_DTYPE) for s in subs]
    for r in range(len(subs)):
        pool.pool_rank(r, streams[r])
    pool.pool_set_photons(list(range(len(subs))), recs)
    return pool


def _same_photons(a, b, what):
    for k in FLOAT_FIELDS + INT_FIELDS:
        assert np.array_equal(a[k], b[k], equal_nan=True), (what, k)


def _setup(hip, lens, lumi=1e54, nzc=8, stokes=1):
    frame, ph, cfg = synth.config2(n_photons=sum(lens), nzc=nzc, stokes=stokes, lumi=lumi)
    subs = _lists(ph, lens)
    R = len(lens)
    streams = [5 + 3 * r for r in range(R)]
    return frame, cfg, subs, streams


def _run_both(hip, lens, F, open_, restore, window, lumi=1e54, monkeypatch=None, cap=None):
    frame, cfg, subs, streams = _setup(hip, lens, lumi=lumi)
    R = len(lens)
    fps = frame["fps"]
    seeds = np.array([[1000 + 17 * r + 1000003 * f for r in range(R)] for f in range(F)], dtype=np.uint64)

    ref = _pool(hip, frame, cfg, subs, streams, window)
    if restore:
        ref.snapshot_photons()
    t_now = [0.0] * R
    want = []
    for f in range(F):
        if restore:
            ref.restore_photons()
        for r in range(R):
            if open_[f][r]:
                t0 = 0.0 if restore else t_now[r]
                rem = 1.0 / fps if restore else (f + 1) / fps - t0
                ref.views[r].begin_frame(int(seeds[f][r]), t0, rem)
        ref.run(0)
        row = []
        for r in range(R):
            if not open_[f][r]:
                row.append(None)
                continue
            st = ref.views[r].frame_statistics()
            t_now[r] = st.time_now
            row.append({k: getattr(st, k) for k in STAT_FIELDS})
        want.append(row)

    q = _pool(hip, frame, cfg, subs, streams, window)
    if restore:
        q.snapshot_photons()
    if cap is not None:
        monkeypatch.setenv("MCRAT_HIP_RANK_LAUNCH_CAP", str(cap))
    t_first = np.zeros((F, R))
    rem_first = np.array([[(1.0 / fps if restore else (f + 1) / fps) for r in range(R)] for f in range(F)])
    frame_end = np.array([[(f + 1) / fps for r in range(R)] for f in range(F)])
    got = q.pool_run_frames(open_, seeds, t_first, rem_first, frame_end=None if restore else frame_end, chain_clock=not restore,
                            restore_each_frame=restore)
    if cap is not None:
        monkeypatch.delenv("MCRAT_HIP_RANK_LAUNCH_CAP")
    for f in range(F):
        for r in range(R):
            if not open_[f][r]:
                assert got[f][r].iterations == 0
                continue
            for k in STAT_FIELDS:
                a, b = getattr(got[f][r], k), want[f][r][k]
                assert a == b or (a != a and b != b), (f, r, k, a, b)
    for r in range(R):
        if any(open_[f][r] for f in range(F)):
            _same_photons(q.views[r].get_photons(), ref.views[r].get_photons(), r)
            st = q.views[r].frame_statistics()             # the view is where the last frame left it
            last = max(f for f in range(F) if open_[f][r])
            assert st.time_now == want[last][r]["time_now"] and st.frame_scatt_cnt == want[last][r]["frame_scatt_cnt"]
    events = sum(w["frame_scatt_cnt"] for row in want for w in row if w)
    assert events > 0
    ref.close()
    q.close()
    return want


FORMS = {"queue": {}, "frame-by-frame": {"MCRAT_HIP_NO_FRAME_QUEUE": "1"}, "128-thread-lists": {"MCRAT_HIP_RANK_BLOCK": "128"},
         "columns-in-hbm": {"MCRAT_HIP_NO_LDS_LISTS": "1"}}


@pytest.mark.parametrize("form", list(FORMS))
def test_chained_frames_equal_one_launch_per_frame(hip, monkeypatch, form):
    """ragged lists, three frames with the clock carried from frame to frame; one list joins at frame 1 (its injection frame).  The queue builds of
    the kernel exist for 256-thread lists with their columns in LDS (kernels.hip); every other launch form runs the plan one launch per frame inside
    the same call -- the same photons either way"""
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    lens = [137, 1000, 512, 999, 64, 700, 1024, 333, 420]
    F, R = 3, len(lens)
    open_ = np.ones((F, R), dtype=np.int32)
    open_[0][4] = 0
    _run_both(hip, lens, F, open_, restore=False, window=1100)


@pytest.mark.parametrize("form", list(FORMS))
def test_restored_frames_equal_restore_plus_launch(hip, monkeypatch, form):
    """the benchmark's shape: every frame from the snapshot with its own seeds"""
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    lens = [400, 1000, 512, 976, 1016, 935]
    F, R = 4, len(lens)
    open_ = np.ones((F, R), dtype=np.int32)
    _run_both(hip, lens, F, open_, restore=True, window=1024)


def test_more_lists_than_the_device_holds(hip):
    """700 short lists over three frames: items are drawn from the ticket by whichever workgroup is free, in frame-major order"""
    lens = [48 + (7 * r) % 33 for r in range(700)]
    F, R = 3, len(lens)
    open_ = np.ones((F, R), dtype=np.int32)
    _run_both(hip, lens, F, open_, restore=False, window=96, lumi=1e55)


def test_a_pass_limit_per_launch_only_costs_launches(hip, monkeypatch):
    """lists whose frame does not finish within the launch's pass limit go on in the next launch, frame by frame, with the same results"""
    lens = [300, 800, 512, 64]
    F, R = 3, len(lens)
    open_ = np.ones((F, R), dtype=np.int32)
    _run_both(hip, lens, F, open_, restore=False, window=1024, monkeypatch=monkeypatch, cap=7)


def test_refusals(hip):
    frame, cfg, subs, streams = _setup(hip, [100, 100])
    q = _pool(hip, frame, cfg, subs, streams, 128)
    F, R = 3, 2
    seeds = np.ones((F, R), dtype=np.uint64)
    z = np.zeros((F, R))
    rem = np.full((F, R), 0.2)
    gap = np.array([[1, 1], [0, 1], [1, 1]], dtype=np.int32)          # a list's frames must be consecutive
    with pytest.raises(hip.McratHipError):
        q.pool_run_frames(gap, seeds, z, rem)
    with pytest.raises(hip.McratHipError):                             # restore without a snapshot
        q.pool_run_frames(np.ones((F, R), dtype=np.int32), seeds, z, rem, restore_each_frame=True)
    with pytest.raises(hip.McratHipError):                             # a chained clock needs the frames' ends
        q.pool_run_frames(np.ones((F, R), dtype=np.int32), seeds, z, rem, chain_clock=True)
    q.close()


def _evolved(frame, k):
    """a later hydro frame of the same mesh: denser, cooler, the flow turned a little (gamma kept consistent with the velocity)"""
    out = dict(frame)
    s = 1.0 - 0.04 * k
    out["v0"] = frame["v0"] * s + 0.01 * k * frame["v1"]
    out["v1"] = frame["v1"] * s
    g = 1.0 / np.sqrt(1.0 - (out["v0"] ** 2 + out["v1"] ** 2))
    out["gamma"] = g
    out["dens"] = frame["dens"] * (1.0 + 0.5 * k)
    out["dens_lab"] = out["dens"] * g
    out["temp"] = frame["temp"] * (1.0 - 0.1 * k)
    out["pres"] = frame["pres"] * (1.0 - 0.1 * k) ** 4
    return out


@pytest.mark.parametrize("form", ["queue", "frame-by-frame"])
def test_every_frame_of_a_plan_in_its_own_hydro_frame(hip, monkeypatch, form):
    """mcrat.c:566-934: a rank reads hydro frame f, runs its photons to the frame's end, reads f + 1.  A plan names the context each frame's hydro
    data is staged on; a list that is through frame f goes on in frame f + 1's cells inside the same launch, while others are still in f.  Three
    frames on three contexts (the pool's own first) against mcrat_hip_share_hydro + one launch per frame"""
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    lens = [137, 1000, 512, 999, 64, 700, 1024, 333, 420]
    frame, cfg, subs, streams = _setup(hip, lens)
    F, R = 3, len(lens)
    fps = frame["fps"]
    frames = [frame, _evolved(frame, 1), _evolved(frame, 2)]
    holders = []
    for f in (1, 2):
        h = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
        h.set_hydro(frames[f])
        holders.append(h)
    seeds = np.array([[77 + 13 * r + 1000003 * f for r in range(R)] for f in range(F)], dtype=np.uint64)
    open_ = np.ones((F, R), dtype=np.int32)
    open_[0][2] = 0                                                     # joins at frame 1

    ref = _pool(hip, frame, cfg, subs, streams, 1100)
    t_now = [0.0] * R
    want = []
    for f in range(F):
        if f:
            ref.share_hydro(holders[f - 1])
        for r in range(R):
            if open_[f][r]:
                ref.views[r].begin_frame(int(seeds[f][r]), t_now[r], (f + 1) / fps - t_now[r])
        ref.run(0)
        row = []
        for r in range(R):
            st = ref.views[r].frame_statistics() if open_[f][r] else None
            if st is not None:
                t_now[r] = st.time_now
            row.append(st and {k: getattr(st, k) for k in STAT_FIELDS})
        want.append(row)
    # the frames differ where it matters: a list's second frame in the first frame's cells is another trajectory
    same = _pool(hip, frame, cfg, subs, streams, 1100)
    frame_end = np.array([[(f + 1) / fps for r in range(R)] for f in range(F)])
    rem = frame_end.copy()
    same.pool_run_frames(open_, seeds, np.zeros((F, R)), rem, frame_end=frame_end, chain_clock=True)
    assert any(not np.array_equal(same.views[r].get_photons()["p0"], ref.views[r].get_photons()["p0"]) for r in range(R))
    same.close()

    q = _pool(hip, frame, cfg, subs, streams, 1100)
    got = q.pool_run_frames(open_, seeds, np.zeros((F, R)), rem, frame_end=frame_end, chain_clock=True, hydro=[None, holders[0], holders[1]])
    for f in range(F):
        for r in range(R):
            if not open_[f][r]:
                continue
            for k in STAT_FIELDS:
                a, b = getattr(got[f][r], k), want[f][r][k]
                assert a == b or (a != a and b != b), (f, r, k, a, b)
    for r in range(R):
        _same_photons(q.views[r].get_photons(), ref.views[r].get_photons(), r)
    assert sum(w["frame_scatt_cnt"] for row in want for w in row if w) > 0

    # a context without a staged frame, or built with other switches, is refused
    empty = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"])
    with pytest.raises(hip.McratHipError):
        q.pool_run_frames(open_, seeds, np.zeros((F, R)), rem, frame_end=frame_end, chain_clock=True, hydro=[None, empty, None])
    other = hip.Engine(synth.TWO, synth.CARTESIAN, cfg["stokes"])
    f1, _, _ = synth.config1(n_photons=10, n0=16, n1=16, stokes=cfg["stokes"])
    other.set_hydro(f1)
    with pytest.raises(hip.McratHipError):
        q.pool_run_frames(open_, seeds, np.zeros((F, R)), rem, frame_end=frame_end, chain_clock=True, hydro=[None, other, None])
    for e in (empty, other, q, ref) + tuple(holders):
        e.close()


@pytest.mark.parametrize("form", ["queue", "frame-by-frame", "128-thread-lists"])
def test_captured_frames_are_the_lists_as_each_frame_left_them(hip, monkeypatch, form):
    """mcrat.c:881-915: phScattStats, saveCheckpoint and printPhotons of frame f need every list as frame f left it -- in a queue launch its next frame
    moves it on at once.  plan.capture_frames keeps a copy per frame; mcrat_hip_pool_select_frame points the pool's read calls at it.  Against the pool
    read after each frame of a frame-by-frame run: records and per-list summaries, bit for bit"""
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    lens = [137, 1000, 512, 999, 64, 700, 1024, 333, 420]
    frame, cfg, subs, streams = _setup(hip, lens)
    F, R, window = 3, len(lens), 1100
    fps = frame["fps"]
    seeds = np.array([[31 + 7 * r + 1000003 * f for r in range(R)] for f in range(F)], dtype=np.uint64)
    open_ = np.ones((F, R), dtype=np.int32)
    ref = _pool(hip, frame, cfg, subs, streams, window)
    t_now = [0.0] * R
    want = []
    for f in range(F):
        for r in range(R):
            ref.views[r].begin_frame(int(seeds[f][r]), t_now[r], (f + 1) / fps - t_now[r])
        ref.run(0)
        for r in range(R):
            t_now[r] = ref.views[r].frame_statistics().time_now
        want.append((ref.get_photons_range(0, R * window).copy(), ref.pool_summaries(), ref.get_output()))
    # (the captures' memory is whatever the allocator hands out: the slots no list owns -- the tails of the windows -- must still read as empty ones)
    import torch
    junk = torch.full((64 << 20,), 3.0e7, dtype=torch.float64, device="cuda")
    del junk
    torch.cuda.empty_cache()
    q = _pool(hip, frame, cfg, subs, streams, window)
    frame_end = np.array([[(f + 1) / fps for r in range(R)] for f in range(F)])
    q.pool_run_frames(open_, seeds, np.zeros((F, R)), frame_end.copy(), frame_end=frame_end, chain_clock=True, capture=True)
    for f in range(F):
        q.pool_select_frame(f if f < F - 1 else -1)
        if f < F - 1:
            with pytest.raises(hip.McratHipError):                                  # nothing runs while a capture is selected
                q.run(0)
        got, summ = q.get_photons_range(0, R * window), q.pool_summaries()
        out = q.get_output()                                                        # printPhotons' compaction over the whole pool: weight != 0
        assert len(out["p0"]) == len(want[f][2]["p0"]) == sum(lens)
        for k in ("p0", "r0", "weight", "num_scatt", "s3"):
            assert np.array_equal(out[k], want[f][2][k]), (f, k)
        for r in range(R):
            a, b = got[r * window:r * window + lens[r]], want[f][0][r * window:r * window + lens[r]]
            for name in a.dtype.names:                                              # (field by field: the records' padding bytes are not data)
                assert np.array_equal(a[name], b[name], equal_nan=a[name].dtype.kind == "f"), (f, r, name)
            for k in ("min_r", "max_r", "min_theta", "max_theta", "avg_r", "avg_scatt", "max_scatt", "min_scatt", "num_output", "list_capacity"):
                x, y = getattr(summ[r], k), getattr(want[f][1][r], k)
                assert x == y or (x != x and y != y), (f, r, k, x, y)
    q.pool_select_frame(-1)
    with pytest.raises(hip.McratHipError):
        q.pool_select_frame(F - 1)                                                  # the last frame is the live lists
    ref.close()
    q.close()


def test_a_plan_with_staged_frames_in_table_mode(hip):
    """TAU_CALCULATION == TABLE: the cross-section table travels with each staged frame's context (its HydroDev carries the table); a context without
    one is refused"""
    from tests.test_gpu_pool import _hot_table
    tab = _hot_table()
    lens = [300, 700, 512]
    frame, cfg, subs, streams = _setup(hip, lens)
    F, R = 2, len(lens)
    fps = frame["fps"]
    later = _evolved(frame, 1)

    def pool_of(fr):
        e = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
        e.set_hot_cross_section(tab)
        e.set_hydro(fr)
        return e

    def filled():
        e = pool_of(frame)
        e.pool_create(R, 768)
        for r in range(R):
            e.pool_rank(r, streams[r])
        e.pool_set_photons(list(range(R)), [synth.photons_to_aos(s, hip.PHOTON_DTYPE) for s in subs])
        return e
    holder = pool_of(later)
    seeds = np.array([[5 + r + 1000003 * f for r in range(R)] for f in range(F)], dtype=np.uint64)
    open_ = np.ones((F, R), dtype=np.int32)
    frame_end = np.array([[(f + 1) / fps for r in range(R)] for f in range(F)])
    ref = filled()
    t_now = [0.0] * R
    for f in range(F):
        if f:
            ref.share_hydro(holder)
        for r in range(R):
            ref.views[r].begin_frame(int(seeds[f][r]), t_now[r], (f + 1) / fps - t_now[r])
        ref.run(0)
        t_now = [ref.views[r].frame_statistics().time_now for r in range(R)]
    q = filled()
    got = q.pool_run_frames(open_, seeds, np.zeros((F, R)), frame_end.copy(), frame_end=frame_end, chain_clock=True, hydro=[None, holder])
    assert sum(got[f][r].frame_scatt_cnt for f in range(F) for r in range(R)) > 0
    for r in range(R):
        _same_photons(q.views[r].get_photons(), ref.views[r].get_photons(), r)
    bare = hip.Engine(cfg["dimensions"], cfg["geometry"], cfg["stokes"], tau_calculation=hip.TAU_TABLE)
    bare.set_hydro(later)                                                           # no table on this one
    with pytest.raises(hip.McratHipError):
        q.pool_run_frames(open_, seeds, np.zeros((F, R)), frame_end.copy(), frame_end=frame_end, chain_clock=True, hydro=[None, bare])
    for e in (bare, q, ref, holder):
        e.close()
