"""Compact score rows (include/ssw_amd.h, round 5): with a plan of the batch's alignments the
scorer stores each utterance's own states' scores only, and the alignment kernels read those rows.
Everything must equal the full-row pipeline bit for bit: the rows themselves (checked against
full rows gathered on the host), and the state alignments through every alignment kernel --
which are checked against the oracle elsewhere (tests/test_gpu_align.py, golden config 3)."""
import json
import os

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import synth_alignment_task, synth_features
from tests.conftest import ROOT
from tests.test_golden_fixtures import crc

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def _task(gpu, orc, means, lens, phones, seed):
    feats = np.concatenate([synth_features(means, n, seed + i) for i, n in enumerate(lens)])
    senid, tmat = [], []
    for i, n in enumerate(phones):
        s, t, _ = synth_alignment_task(orc.sseq, orc.phone_ssid, orc.phone_tmat, orc.n_ciphone, n,
                                       seed + 100 + i)
        senid.append(s)
        tmat.append(t)
    frame_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum(phones)]).astype(np.int32)
    return feats, frame_off, phone_off, np.concatenate(senid), np.concatenate(tmat)


def _compact_rows(gpu, plan, d_c, frame_off, phone_off):
    buf = np.zeros(plan.elems, np.int16)
    gpu._L.ssw_memcpy_d2h(buf.ctypes.data, d_c, buf.nbytes)
    out = []
    for u in range(len(frame_off) - 1):
        off, stride = plan.rows(u)
        nf = frame_off[u + 1] - frame_off[u]
        out.append(buf[off:off + nf * stride].reshape(nf, stride))
    return out


def _both_ways(gpu, feats, frame_off, phone_off, senid, tmat, sf=None, ef=None,
               scorer=ssw.SCORER_PTM):
    d_feats = gpu.to_device(feats)
    d_full = gpu.device_malloc(len(feats) * gpu.n_sen * 2)
    plan = gpu.compact_plan(frame_off, phone_off, senid)
    d_c = gpu.device_malloc(max(plan.nbytes, 2))
    try:
        gpu.score_batch_device(d_feats, len(feats), frame_off, d_full, scorer=scorer)
        full = np.zeros((len(feats), gpu.n_sen), np.int16)
        gpu._L.ssw_device_synchronize()
        gpu._L.ssw_memcpy_d2h(full.ctypes.data, d_full, full.nbytes)
        st_f, status_f = gpu.align_batch(d_full, frame_off, phone_off, senid, tmat, sf, ef)
        gpu.score_batch_compact(d_feats, plan, d_c, scorer=scorer)
        gpu._L.ssw_device_synchronize()
        rows = _compact_rows(gpu, plan, d_c, frame_off, phone_off)
        # output only (SSW_ALIGN_STATE_OUT_ONLY): whatever the array held must not matter
        junk = np.full((len(tmat) * 3, 3), 0x5a5a5a5a, np.int32)
        st_c, status_c = gpu.align_batch_compact(plan, d_c, tmat, sf, ef, out=junk)
    finally:
        for p in (d_feats, d_full, d_c):
            gpu.device_free(p)
        plan.free()
    # the rows: state k's column holds its senone's score -- at the first state with that senone
    sen = np.asarray(senid, np.uint16).reshape(-1)
    for u, r in enumerate(rows):
        a, b = frame_off[u], frame_off[u + 1]
        ids = sen[phone_off[u] * 3:phone_off[u + 1] * 3].astype(np.int64)
        first = {}
        for k, s_ in enumerate(ids):
            first.setdefault(int(s_), k)
        cols = np.array([first[int(s_)] for s_ in ids])
        assert np.array_equal(r[:, cols], full[a:b][:, ids]), ("rows", u)
    assert np.array_equal(status_c, status_f)
    assert np.array_equal(st_c, st_f)
    return st_c, status_c


def test_compact_rows_and_alignments_equal_the_full_row_pipeline(gpu_en, orc_en, means_en):
    lens = [300, 41, 1, 120, 77, 256]
    phones = [40, 9, 1, 30, 150, 64]          # 150 phones: three waves; 1 phone in 1 frame
    task = _task(gpu_en, orc_en, means_en, lens, phones, 900)
    st, status = _both_ways(gpu_en, *task)
    assert (status == 0).sum() >= 4


def test_compact_rows_in_a_batch_large_enough_for_the_persistent_kernel_and_pieces(
        gpu_en, orc_en, means_en, monkeypatch):
    lens = [257] * 11 + [120, 3]              # 2,950 frames: two frames per group; ragged tail
    phones = [38] * 11 + [20, 1]
    task = _task(gpu_en, orc_en, means_en, lens, phones, 1200)
    ref = _both_ways(gpu_en, *task)
    for knobs in ({"SSW_SCORE_PIECE": "768"}, {"SSW_SEN_FPB": "1"}, {"SSW_SEN_R": "4"},
                  {"SSW_SEN_GENERIC": "1"}, {"SSW_SEN_GROUPS": "0"}, {"SSW_SCAN": "fma"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        got = _both_ways(gpu_en, *task)
        for k in knobs:
            monkeypatch.delenv(k)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), knobs


@pytest.mark.parametrize("kernel", ["mw", "reg", "lds", "hbm", "win"])
def test_every_alignment_kernel_reads_compact_rows(gpu_en, orc_en, means_en, monkeypatch, kernel):
    lens = [200, 90, 150]
    phones = [60, 20, 70]
    feats, frame_off, phone_off, senid, tmat = _task(gpu_en, orc_en, means_en, lens, phones, 1500)
    # word-like windows for one of them
    sf = np.zeros(len(tmat), np.int32)
    ef = np.full(len(tmat), INT_MAX, np.int32)
    p0, p1 = phone_off[2], phone_off[3]
    per = lens[2] / phones[2]
    for i in range(p0, p1):
        c = int((i - p0) * per)
        sf[i], ef[i] = max(0, c - 25), min(lens[2], c + 40)
    sf[p0:p1] = np.maximum.accumulate(sf[p0:p1])
    ef[p0:p1] = np.maximum.accumulate(ef[p0:p1])
    ref = _both_ways(gpu_en, feats, frame_off, phone_off, senid, tmat, sf, ef)
    monkeypatch.setenv("SSW_ALIGN_KERNEL", kernel)
    got = _both_ways(gpu_en, feats, frame_off, phone_off, senid, tmat, sf, ef)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_output_only_entries_survive_a_hand_on_to_the_full_token_kernel(gpu_en, orc_en, means_en):
    """with SSW_ALIGN_STATE_OUT_ONLY (the default of align_batch_compact, of config 5
    and of bench.py's config 3) a batch of which the byte-token kernel hands some utterances on is
    run twice -- and the second run used to clear the device's entries and copy ALL of them back,
    zeros for every utterance the first run had aligned, status 0.  Windows that end a phone
    while its predecessor lives on (tests/test_gpu_align.py::test_byte_token_kernel_hands_on...)
    make the hand-on happen; the entries must equal the full-row in / out call's, which that
    test compares with the oracle."""
    rng = np.random.default_rng(123)
    n_utts = 24
    phones = rng.integers(4, 140, size=n_utts).tolist()
    lens = [int(p * rng.integers(3, 6) + 6) for p in phones]
    feats, frame_off, phone_off, senid, tmat = _task(gpu_en, orc_en, means_en, lens, phones, 2100)
    sf, ef = [], []
    for p, f in zip(phones, lens):
        mid_ = (np.arange(p) * f) // p
        a = np.maximum(mid_ - 8, 0).astype(np.int32)
        b = np.minimum(mid_ + f // p + 10, f).astype(np.int32)
        b[1::3] = np.maximum(a[1::3] + 2, mid_[1::3] - 2)
        b[-1] = f
        sf.append(a)
        ef.append(b)
    sf, ef = np.concatenate(sf), np.concatenate(ef)
    before = gpu_en.align_stats()
    st, status = _both_ways(gpu_en, feats, frame_off, phone_off, senid, tmat, sf, ef)
    after = gpu_en.align_stats()
    assert after[1] - before[1] >= 2, "the windows were meant to make the byte-token kernel hand on"
    ok = status == 0
    assert 4 <= ok.sum(), status
    for u in np.flatnonzero(ok):              # aligned utterances tile their frames
        s = st[phone_off[u] * 3:phone_off[u + 1] * 3]
        assert s[0, 0] == 0 and s[:, 1].sum() == lens[u], u
    # against the oracle, from compact-call entries alone
    d_feats = gpu_en.to_device(feats)
    d_full = gpu_en.device_malloc(len(feats) * gpu_en.n_sen * 2)
    try:
        gpu_en.score_batch_device(d_feats, len(feats), frame_off, d_full)
        full = np.zeros((len(feats), gpu_en.n_sen), np.int16)
        gpu_en._L.ssw_device_synchronize()
        gpu_en._L.ssw_memcpy_d2h(full.ctypes.data, d_full, full.nbytes)
    finally:
        gpu_en.device_free(d_feats)
        gpu_en.device_free(d_full)
    for u in range(n_utts):
        sl = slice(phone_off[u], phone_off[u + 1])
        rv, rst, _ = orc_en.state_align(full[frame_off[u]:frame_off[u + 1]], senid[sl], tmat[sl],
                                        sf=sf[sl], ef=ef[sl])
        assert (status[u] == 0) == (rv == 0), u
        if rv == 0:
            assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), u


def test_output_only_entries_survive_a_fall_back_one_level_down(gpu_en, orc_en, means_en,
                                                                 monkeypatch):
    """The other rerun: an utterance that outgrows the sliding-window kernel (-(1 << 30)) is run
    again a level down; with output-only entries the rerun starts from zeros, like the first run,
    and the utterances the first run aligned keep their entries."""
    shapes = [(300, 950), (300, 950), (5, 3), (192, 600), (40, 130), (192, 600), (8, 2)]
    windowed = [True, False, False, False, False, True, False]
    phones = [a for a, _ in shapes]
    lens = [b for _, b in shapes]
    feats, frame_off, phone_off, senid, tmat = _task(gpu_en, orc_en, means_en, lens, phones, 2300)
    sf, ef = [], []
    for (p, f), w in zip(shapes, windowed):
        a = np.zeros(p, np.int32)
        b = np.full(p, INT_MAX, np.int32)
        if w:
            mid = (np.arange(p) * f) // p
            a = np.maximum(mid - 5, 0).astype(np.int32)
            b = np.minimum(mid + f // p + 7, f).astype(np.int32)
        sf.append(a)
        ef.append(b)
    sf, ef = np.concatenate(sf), np.concatenate(ef)
    ref = _both_ways(gpu_en, feats, frame_off, phone_off, senid, tmat, sf, ef)
    monkeypatch.setenv("SSW_ALIGN_KERNEL", "win")
    monkeypatch.setenv("SSW_ALIGN_WIN_WAVES", "2")
    got = _both_ways(gpu_en, feats, frame_off, phone_off, senid, tmat, sf, ef)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert (got[1] == 0).sum() >= 5 and (got[1] != 0).sum() >= 1
    assert not (got[1] == -(1 << 30)).any()


def test_golden_config3_through_compact_rows(gpu_en, orc_en, means_en):
    """the four reference-confirmed state alignments of BASELINE config 3"""
    with open(os.path.join(ROOT, "tests", "golden", "synthetic_oracle.json")) as fh:
        golden = json.load(fh)
    feats = np.concatenate([synth_features(means_en, 1000, 12345 + u) for u in range(4)])
    frame_off = (np.arange(5) * 1000).astype(np.int32)
    phone_off = (np.arange(5) * 150).astype(np.int32)
    senid, tmat = [], []
    for u in range(4):
        s, t, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                       orc_en.n_ciphone, 150, 777 + u)
        senid.append(s)
        tmat.append(t)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    plan = gpu_en.compact_plan(frame_off, phone_off, senid)
    d_feats = gpu_en.to_device(feats)
    d_c = gpu_en.device_malloc(plan.nbytes)
    try:
        assert plan.nbytes == 4 * 1000 * 450 * 2
        gpu_en.score_batch_compact(d_feats, plan, d_c)
        st, status = gpu_en.align_batch_compact(plan, d_c, tmat)
    finally:
        gpu_en.device_free(d_feats)
        gpu_en.device_free(d_c)
        plan.free()
    for u, g in enumerate(golden["config3_align"]):
        assert (status[u] == 0) == (g["rv"] == 0)
        assert crc(st[u * 450:(u + 1) * 450]) == g["states_crc"]


def test_ms_scorer_compact_rows_direct_and_through_the_fall_back(monkeypatch):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ms
    from oracle import oracle as O
    m, means = bench_ms.build_model()
    try:
        class _T:      # the tables synth_alignment_task needs, from the GPU model
            sseq = m.table("sseq").reshape(-1, 3)
            phone_ssid, phone_tmat, n_ciphone = m.table("phone_ssid"), m.table("phone_tmat"), m.n_ciphone
        lens = [400] * 6 + [33]               # 2,433 frames: the packed ms kernel, direct
        phones = [50] * 6 + [5]
        task = _task(m, _T, means, lens, phones, 2100)
        a = _both_ways(m, *task, scorer=ssw.SCORER_MS)
        monkeypatch.setenv("SSW_MS_SENONE", "old")        # no COMPACT instance: gather fall-back
        b = _both_ways(m, *task, scorer=ssw.SCORER_MS)
        monkeypatch.delenv("SSW_MS_SENONE")
        assert np.array_equal(a[0], b[0])
        small = _task(m, _T, means, [120, 60], [20, 11], 2200)   # small batch: fall-back too
        _both_ways(m, *small, scorer=ssw.SCORER_MS)
    finally:
        m.close()


def test_plan_refuses_what_it_cannot_hold(gpu_en):
    with pytest.raises(ssw.SswError, match="more than a compact row holds"):
        gpu_en.compact_plan([0, 10], [0, 11000], np.zeros((11000, 3), np.uint16))
    with pytest.raises(ssw.SswError, match="senone"):
        gpu_en.compact_plan([0, 10], [0, 2], np.full((2, 3), 60000, np.uint16))
