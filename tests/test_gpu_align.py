"""GPU parity: forced-alignment Viterbi through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from soundswallower_amd.synth import lcg_uniform, synth_alignment_task

pytestmark = pytest.mark.gpu


def _random_senscr(n_frames, n_sen, seed):
    u = lcg_uniform(seed, n_frames * n_sen).reshape(n_frames, n_sen)
    scr = np.floor(u * 600).astype(np.int16)
    scr[np.arange(n_frames), np.floor(lcg_uniform(seed + 1, n_frames) * n_sen).astype(int)] = 0
    return scr


@pytest.mark.parametrize("n_phones,n_frames", [(1, 5), (5, 40), (64, 300), (65, 300), (128, 500), (129, 500), (150, 700), (256, 900),
                          (257, 900), (513, 1700)])
def test_align_matches_oracle(gpu_en, orc_en, n_phones, n_frames):
    senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                          orc_en.n_ciphone, n_phones, 99 + n_phones)
    scr = _random_senscr(n_frames, orc_en.n_sen, 7 + n_phones)
    rv, rst, rph = orc_en.state_align(scr, senid, tmat)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat)
    finally:
        gpu_en.device_free(d)
    assert (status[0] == 0) == (rv == 0)
    if rv == 0:
        assert np.array_equal(st, rst)


def test_all_alignment_kernels_agree_on_a_ragged_windowed_batch(gpu_en, orc_en, monkeypatch):
    """Utterances of 1..200 phones in one call, with phone windows (sf/ef) on some of them and
    one window that cannot be met: the wave-per-word kernel (the default here), the
    wave-per-utterance register kernel, the LDS kernel, the HBM-resident one and the
    sliding-window one (SSW_ALIGN_KERNEL=mw/reg/lds/hbm/win) must all equal the oracle, failures
    included.  (The utterances without windows enter every phone in frame 0: the window kernel
    reports that it cannot hold them and the call falls back, which is part of what is tested.)"""
    rng = np.random.default_rng(5)
    n_ph = [1, 3, 64, 65, 127, 128, 200, 17]
    n_fr = [int(p * rng.integers(3, 6) + 4) for p in n_ph]
    frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
    scr = _random_senscr(int(frame_off[-1]), orc_en.n_sen, 31)
    senid, tmat, sf, ef = [], [], [], []
    for u, (p, f) in enumerate(zip(n_ph, n_fr)):
        s, t, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                       orc_en.n_ciphone, p, 400 + u)
        senid.append(s)
        tmat.append(t)
        a = np.zeros(p, np.int32)
        b = np.full(p, 2**31 - 1, np.int32)
        if u % 2 == 1:  # loose windows around an even split
            mid = (np.arange(p) * f) // p
            a = np.maximum(mid - 6, 0).astype(np.int32)
            b = np.minimum(mid + f // p + 8, f).astype(np.int32)
        if u == len(n_ph) - 1:  # phone 3 may not start before the last frame: no path
            a[3:] = f
        sf.append(a)
        ef.append(b)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    sf, ef = np.concatenate(sf), np.concatenate(ef)
    ref = []
    for u in range(len(n_ph)):
        sl = slice(phone_off[u], phone_off[u + 1])
        ref.append(orc_en.state_align(scr[frame_off[u]:frame_off[u + 1]], senid[sl], tmat[sl],
                                      sf=sf[sl], ef=ef[sl]))
    d = gpu_en.to_device(scr)
    try:
        results = []
        monkeypatch.setenv("SSW_ALIGN_WIN_WAVES", "2")   # "win": a window of 2 blocks that slides
        for mode in ("mw", "reg", "lds", "hbm", "win", "mwb"):
            monkeypatch.setenv("SSW_ALIGN_KERNEL", mode)
            results.append(gpu_en.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef))
    finally:
        gpu_en.device_free(d)
    assert any(r[0] != 0 for r in ref) and any(r[0] == 0 for r in ref)
    for st, status in results:
        for u, (rv, rst, _) in enumerate(ref):
            assert (status[u] == 0) == (rv == 0), u
            if rv == 0:
                assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), u
    assert np.array_equal(results[0][1], results[1][1])
    assert np.array_equal(results[0][1], results[2][1])
    assert np.array_equal(results[0][1], results[3][1])
    assert np.array_equal(results[0][1], results[4][1])
    assert np.array_equal(results[0][1], results[5][1])


@pytest.mark.parametrize("mode", ["mw", "reg", "lds", "hbm"])
def test_other_alignment_kernels_match_oracle(gpu_en, orc_en, monkeypatch, mode):
    monkeypatch.setenv("SSW_ALIGN_KERNEL", mode)
    for n_phones, n_frames in ((64, 300), (65, 300), (150, 700), (256, 900)):
        test_align_matches_oracle(gpu_en, orc_en, n_phones, n_frames)


def test_skip_arc_topologies_match_oracle(orc_en, oracle_mod, monkeypatch, tmp_path):
    """Both shipped models have no skip arcs, so the skip branches of hmm_vit_eval_3st_lr
    (src/hmm.c:496-535, including the t2 left over from the exit block) never run on them.
    A synthetic transition_matrices file with 0->2 and 1->exit arcs in some matrices and not in
    others exercises them in all three alignment kernels."""
    import os
    import struct

    import soundswallower_amd as ssw
    from tests.test_cabi_host import _write_s3

    rng = np.random.default_rng(77)
    n_tmat = int(orc_en.n_tmat)
    tm = np.zeros((n_tmat, 3, 4), "<f4")
    for i in range(n_tmat):
        kind = i % 4  # 0: both skips, 1: only 0->2, 2: only 1->exit, 3: none
        for j in range(3):
            row = np.zeros(4)
            row[j] = rng.uniform(0.3, 0.8)
            row[j + 1] = rng.uniform(0.2, 0.6)
            if j + 2 <= 3 and ((j == 0 and kind in (0, 1)) or (j == 1 and kind in (0, 2))):
                row[j + 2] = rng.uniform(0.05, 0.5)
            tm[i, j] = row
    path = str(tmp_path / "transition_matrices")
    _write_s3(path, struct.pack("<4i", n_tmat, 3, 4, tm.size) + tm.tobytes())
    src = ssw.model_dir("en-us")
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
              sendump=os.path.join(src, "sendump"), tmat=path)
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    assert np.array_equal(g.table("tp"), o.tp.reshape(-1))
    tp = o.tp
    assert (tp[:, 0, 2] < 255).any() and (tp[:, 0, 2] == 255).any() and (tp[:, 1, 3] < 255).any()
    for mode in ("mw", "reg", "lds", "hbm", "mwb"):
        monkeypatch.setenv("SSW_ALIGN_KERNEL", mode)
        for n_phones, n_frames in ((6, 30), (70, 260), (150, 500)):
            senid, tmat, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone,
                                                  n_phones, 1234 + n_phones)
            scr = _random_senscr(n_frames, o.n_sen, 55 + n_phones)
            rv, rst, _ = o.state_align(scr, senid, tmat)
            d = g.to_device(scr)
            try:
                st, status = g.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat)
            finally:
                g.device_free(d)
            assert (status[0] == 0) == (rv == 0), (mode, n_phones)
            assert rv == 0
            assert np.array_equal(st, rst), (mode, n_phones)


@pytest.mark.timeout(600)
def test_utterance_beyond_the_lds_limit(gpu_en, orc_en):
    """3,000 phones (the LDS kernel holds 2,560): the HBM-resident kernel by itself, with windows
    that keep the walked range short, equal to the oracle."""
    n_phones, n_frames = 3000, 9500
    senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                          orc_en.n_ciphone, n_phones, 4242)
    scr = _random_senscr(n_frames, orc_en.n_sen, 99)
    mid = (np.arange(n_phones) * n_frames) // n_phones
    sf = np.maximum(mid - 40, 0).astype(np.int32)
    ef = np.minimum(mid + 60, n_frames).astype(np.int32)
    rv, rst, _ = orc_en.state_align(scr, senid, tmat, sf=sf, ef=ef)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat, sf=sf, ef=ef)
        # ... and without windows: every phone from 0 to the frontier stays in the walked range
        rv2, rst2, _ = orc_en.state_align(scr[:7000], senid[:2600], tmat[:2600])
        st2, status2 = gpu_en.align_batch(d, [0, 7000], [0, 2600], senid[:2600], tmat[:2600])
        # round 3: the token table of this kernel is banded (the walked 64-phone words of every
        # frame only).  The full [frame][state] table (SSW_ALIGN_BAND=0) and a budget so small
        # that the bands overflow it and the call falls back to the full table
        # (SSW_ALIGN_BAND_TOKENS=192: one word per frame) must give the same entries.
        import os
        variants = []
        for env in ({"SSW_ALIGN_BAND": "0"}, {"SSW_ALIGN_BAND_TOKENS": "192"}):
            os.environ.update(env)
            try:
                variants.append(gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat,
                                                   sf=sf, ef=ef))
                variants.append(gpu_en.align_batch(d, [0, 7000], [0, 2600], senid[:2600], tmat[:2600]))
            finally:
                for k in env:
                    del os.environ[k]
    finally:
        gpu_en.device_free(d)
    assert rv == 0 and status[0] == 0 and np.array_equal(st, rst)
    assert (status2[0] == 0) == (rv2 == 0)
    if rv2 == 0:
        assert np.array_equal(st2, rst2)
    for k, (stv, statusv) in enumerate(variants):
        want_st, want_status = (st, status) if k % 2 == 0 else (st2, status2)
        assert statusv[0] == want_status[0], k
        if want_status[0] == 0:
            assert np.array_equal(stv, want_st), k


@pytest.mark.parametrize("waves", [2, 4, 8, 16])
def test_sliding_window_kernel(gpu_en, orc_en, monkeypatch, waves):
    """Round 3: viterbi_align_win_kernel -- HMMs in registers for utterances of any length, `waves`
    64-phone blocks at a time, the window following the active phones.  700 phones (11 blocks)
    over 2,400 frames with the kind of windows a first pass leaves (every phone confined to a
    stretch around its share of the audio): windows of 2 .. 16 blocks, i.e. from "slides every few
    frames and sometimes cannot hold the active range" (falls back) to "never slides"; all equal
    to the oracle.  Then the same with a token budget of one block per frame (falls back)."""
    n_phones, n_frames = 700, 2400
    senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                          orc_en.n_ciphone, n_phones, 31337)
    scr = _random_senscr(n_frames, orc_en.n_sen, 4711)
    mid = (np.arange(n_phones) * n_frames) // n_phones
    res = {}
    for name, (lo, hi) in {"narrow": (12, 20), "wide": (150, 260)}.items():
        sf = np.maximum(mid - lo, 0).astype(np.int32)
        ef = np.minimum(mid + hi, n_frames).astype(np.int32)
        rv, rst, _ = orc_en.state_align(scr, senid, tmat, sf=sf, ef=ef)
        res[name] = (sf, ef, rv, rst)
    monkeypatch.setenv("SSW_ALIGN_KERNEL", "win")
    monkeypatch.setenv("SSW_ALIGN_WIN_WAVES", str(waves))
    d = gpu_en.to_device(scr)
    try:
        for name, (sf, ef, rv, rst) in res.items():
            st, status = gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat, sf=sf, ef=ef)
            assert (status[0] == 0) == (rv == 0), name
            if rv == 0:
                assert np.array_equal(st, rst), name
        monkeypatch.setenv("SSW_ALIGN_BAND_TOKENS", "192")
        sf, ef, rv, rst = res["narrow"]
        st, status = gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat, sf=sf, ef=ef)
        assert (status[0] == 0) == (rv == 0)
        if rv == 0:
            assert np.array_equal(st, rst)
    finally:
        gpu_en.device_free(d)
    assert res["narrow"][2] == 0          # the narrow windows do leave a path


@pytest.mark.parametrize("mode", ["mw", "win"])
def test_edge_shapes_of_the_register_kernels(gpu_en, orc_en, monkeypatch, mode):
    """Phone counts at the 64-phone block boundaries, utterances of 0, 1 and 2 frames, utterances
    of very different lengths in one call -- through the wave-per-word kernel and through the
    sliding-window one (window of 2 blocks: it slides, and gives up on the long unwindowed
    ones); status and entries equal to the oracle for every utterance."""
    monkeypatch.setenv("SSW_ALIGN_KERNEL", mode)
    monkeypatch.setenv("SSW_ALIGN_WIN_WAVES", "2")
    shapes = [(1, 0), (1, 1), (2, 1), (1, 2), (3, 2), (63, 200), (64, 200), (65, 210), (127, 400),
              (128, 400), (129, 420), (192, 600), (5, 3), (300, 950)]
    n_ph = [a for a, _ in shapes]
    n_fr = [b for _, b in shapes]
    frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
    scr = _random_senscr(int(frame_off[-1]), orc_en.n_sen, 2718)
    senid, tmat, sf, ef = [], [], [], []
    for u, (p, f) in enumerate(shapes):
        s_, t_, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                         orc_en.n_ciphone, p, 900 + u)
        senid.append(s_)
        tmat.append(t_)
        a = np.zeros(p, np.int32)
        b = np.full(p, 2**31 - 1, np.int32)
        if p >= 63 and u % 2 == 1:      # windows on every other long one
            mid = (np.arange(p) * f) // p
            a = np.maximum(mid - 5, 0).astype(np.int32)
            b = np.minimum(mid + f // p + 7, f).astype(np.int32)
        sf.append(a)
        ef.append(b)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    sf, ef = np.concatenate(sf), np.concatenate(ef)
    d = gpu_en.to_device(scr) if len(scr) else None
    try:
        st, status = gpu_en.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef)
    finally:
        gpu_en.device_free(d)
    n_ok = 0
    for u in range(len(shapes)):
        sl = slice(phone_off[u], phone_off[u + 1])
        rv, rst, _ = orc_en.state_align(scr[frame_off[u]:frame_off[u + 1]], senid[sl], tmat[sl],
                                        sf=sf[sl], ef=ef[sl])
        assert (status[u] == 0) == (rv == 0), (u, shapes[u], status[u], rv)
        if rv == 0:
            n_ok += 1
            assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), (u, shapes[u])
    assert n_ok >= 8


def test_fall_back_reruns_only_what_failed_from_the_callers_state(gpu_en, orc_en, monkeypatch):
    """when an utterance outgrows a level (-(1 << 30)) only that utterance is run
    again one level down, from the state entries the CALLER passed in (state_io is in / out),
    not from whatever the failed level copied back.  A batch through the sliding-window kernel
    with a window of two blocks: long utterances with narrow phone windows (stay), long ones
    without (overflow, fall back), short ones, one without a path (too few frames) -- every one
    starts from its own non-trivial entries (start / duration as alignment_populate would leave
    them, a marker in the scores) and must come back as the oracle's state_align_search leaves
    them: the aligned ones rewritten, the failing one's entries untouched."""
    monkeypatch.setenv("SSW_ALIGN_KERNEL", "win")
    monkeypatch.setenv("SSW_ALIGN_WIN_WAVES", "2")
    shapes = [(300, 950), (300, 950), (5, 3), (192, 600), (40, 130), (192, 600), (8, 2)]
    windowed = [True, False, False, False, False, True, False]
    n_ph = [a for a, _ in shapes]
    n_fr = [b for _, b in shapes]
    frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
    scr = _random_senscr(int(frame_off[-1]), orc_en.n_sen, 1618)
    rng = np.random.default_rng(3)
    senid, tmat, sf, ef, init = [], [], [], [], []
    for u, (p, f) in enumerate(shapes):
        s_, t_, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                         orc_en.n_ciphone, p, 1200 + u)
        senid.append(s_)
        tmat.append(t_)
        a = np.zeros(p, np.int32)
        b = np.full(p, 2**31 - 1, np.int32)
        if windowed[u]:
            mid = (np.arange(p) * f) // p
            a = np.maximum(mid - 5, 0).astype(np.int32)
            b = np.minimum(mid + f // p + 7, f).astype(np.int32)
        sf.append(a)
        ef.append(b)
        init.append(np.stack([rng.integers(0, max(f, 1), 3 * p), rng.integers(1, 9, 3 * p),
                              rng.integers(-999, -1, 3 * p)], 1).astype(np.int32))
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    sf, ef, init = np.concatenate(sf), np.concatenate(ef), np.concatenate(init)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef,
                                        state_init=init)
    finally:
        gpu_en.device_free(d)
    n_ok = n_bad = 0
    for u in range(len(shapes)):
        sl = slice(phone_off[u], phone_off[u + 1])
        s3 = slice(phone_off[u] * 3, phone_off[u + 1] * 3)
        rv, rst, _ = orc_en.state_align(scr[frame_off[u]:frame_off[u + 1]], senid[sl], tmat[sl],
                                        sf=sf[sl], ef=ef[sl], state_init=init[s3])
        assert (status[u] == 0) == (rv == 0), (u, shapes[u], status[u], rv)
        assert status[u] != -(1 << 30)
        if rv == 0:
            n_ok += 1
            assert np.array_equal(st[s3], rst), (u, shapes[u])
        else:
            n_bad += 1
            assert np.array_equal(st[s3], init[s3]), (u, shapes[u])   # untouched
    assert n_ok >= 5 and n_bad >= 1


def test_byte_token_kernel_hands_on_what_it_cannot_replay(gpu_en, orc_en):
    """viterbi_align_mwb_kernel (2-bit back-pointers + score replay, the default up to 1024 phones)
    must give the reference's entries or hand the utterance, untouched, to the full-token kernel.
    Ordinary problems: nothing handed on.  Windows that END a phone while its predecessor lives
    on, so that it is entered again with states 1 and 2 still holding their old scores
    (hmm_enter only sets state 0, src/hmm.c:142-148): handed on, and still equal to the oracle."""
    rng = np.random.default_rng(123)
    before = gpu_en.align_stats()
    test_align_matches_oracle(gpu_en, orc_en, 150, 700)
    mid = gpu_en.align_stats()
    assert mid[0] == before[0] + 1 and mid[1] == before[1]
    n_utts, seen_ok = 24, 0
    n_ph = rng.integers(4, 140, size=n_utts).tolist()
    n_fr = [int(p * rng.integers(3, 6) + 6) for p in n_ph]
    frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
    scr = _random_senscr(int(frame_off[-1]), orc_en.n_sen, 99)
    senid, tmat, sf, ef = [], [], [], []
    for u, (p, f) in enumerate(zip(n_ph, n_fr)):
        s_, t_, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                         orc_en.n_ciphone, p, 4000 + u)
        senid.append(s_)
        tmat.append(t_)
        mid_ = (np.arange(p) * f) // p
        a = np.maximum(mid_ - 8, 0).astype(np.int32)
        b = np.minimum(mid_ + f // p + 10, f).astype(np.int32)
        # every third phone's window closes early while its neighbours' stay open: it drops out
        # and is entered again (and again) from its predecessor
        b[1::3] = np.maximum(a[1::3] + 2, mid_[1::3] - 2)
        b[-1] = f
        sf.append(a)
        ef.append(b)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    sf, ef = np.concatenate(sf), np.concatenate(ef)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef)
    finally:
        gpu_en.device_free(d)
    after = gpu_en.align_stats()
    assert after[0] == mid[0] + n_utts
    for u in range(n_utts):
        sl = slice(phone_off[u], phone_off[u + 1])
        rv, rst, _ = orc_en.state_align(scr[frame_off[u]:frame_off[u + 1]], senid[sl], tmat[sl],
                                        sf=sf[sl], ef=ef[sl])
        assert (status[u] == 0) == (rv == 0), u
        if rv == 0:
            seen_ok += 1
            assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), u
    print("handed on to the full-token kernel: %d of %d (aligned: %d)"
          % (after[1] - mid[1], n_utts, seen_ok))
    assert after[1] > mid[1], "the windows were meant to make some utterances irregular"
