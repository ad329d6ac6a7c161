"""Non-finite feature rows: NaN and +-Inf against the oracle compiled on x86
here, which restates the reference's staged dimension walk (src/ptm_mgau.c:150-225) and takes its
(int32) casts from the host CPU as the reference does -- (int32)NaN = 0x80000000.

What the reference does with a NaN feature depends on WHERE it sits: eval_cb leaves a density's
dimension walk as soon as `d >= thresh` fails at the head of a stage (13 dimensions: one single
dimension, then three groups of four), which it does for a NaN d, so a NaN in dimensions 0..8
drops every density of that stream ("terminated early, so not in topn"), while a NaN confined to
dimensions 9..12 lets every density whose partial sum still clears the threshold through to be
inserted with INT32_MIN.  An infinity makes d = -Inf, which every comparison handles.  PTM scorer:
all of that must come out of the GPU bit for bit (scores and top-N codeword order).  ms scorer:
+-Inf only -- with a NaN the reference's compute_dist leaves its top-N ids untouched, i.e. reads
the PREVIOUS frame's (src/ms_gauden.c:397-420, the buffer of src/ms_mgau.c:152), which no batch
can reproduce; include/ssw_amd.h says so."""
import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import synth_features
from tests.test_gpu_ptm import _oracle_batch

pytestmark = pytest.mark.gpu
NAN, INF = np.float32(np.nan), np.float32(np.inf)


def _poisoned(means, seed=99):
    lens = [70, 33, 64, 1, 90]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    f = np.concatenate([synth_features(means, n, seed + i) for i, n in enumerate(lens)])
    rng = np.random.default_rng(seed)
    marks = []

    def put(t, dims, v):
        for d in np.atleast_1d(dims):
            f[t, d] = v
        marks.append(t)

    # one NaN per stage of the walk, in each stream (stream s = columns 13 s .. 13 s + 12)
    t = 3
    for s_ in range(3):
        for d in (0, 1, 4, 5, 8, 9, 10, 12):
            put(t, 13 * s_ + d, NAN)
            t += 2
    put(60, [9, 10, 11, 12], NAN)               # the whole last stage
    put(62, [12, 13 + 12, 26 + 12], NAN)         # late in every stream
    put(63, [12, 13 + 3], NAN)                   # late in one stream, early in another
    put(64, np.arange(39), NAN)                  # a row of NaNs
    put(65, np.arange(39), NAN)                  # ... twice (the carried list is all INT_MIN)
    put(66, [9], NAN)                            # late, from an all-INT_MIN history
    put(70, [12], NAN)                           # first frame of utterance 1: reset history
    put(int(off[2]) - 1, [11], NAN)              # last frame of utterance 1
    put(int(off[3]), [10], NAN)                  # the one-frame utterance
    # infinities, alone and with NaNs
    put(110, [0], INF)
    put(111, [7], -INF)
    put(112, [12], INF)
    put(113, [2, 30], INF)
    put(114, np.arange(13), -INF)
    put(115, [3], INF); f[115, 11] = NAN         # -Inf before the last stage, NaN in it
    put(116, [11], INF); f[116, 2] = NAN         # NaN early
    # late NaNs in runs (each frame starts from the list the one before left)
    for t in range(180, 200):
        put(t, [13 + 9 + int(rng.integers(0, 4))], NAN)
    # huge but finite: d below the int range without being -Inf
    put(210, [5], np.float32(3.0e19))
    put(211, [12], np.float32(-1.0e30))
    return f, off, sorted(set(marks))


def _check(gpu, orc, feats, off, what):
    got = gpu.score_batch(feats, off)
    gcw, _ = gpu.last_topn(len(feats))         # (its scores are the raw ones; the oracle's view
    ref, rcw, _ = _oracle_batch(orc, feats, off)   # holds them normalised: codewords compared)
    bad = np.nonzero((gcw.astype(np.int32) != rcw).any(axis=(1, 2, 3)))[0]
    assert len(bad) == 0, (what, "top-N codewords", bad[:10].tolist())
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert len(bad) == 0, (what, "senone scores", bad[:10].tolist())


def test_ptm_nan_and_inf_rows_match_the_reference_arithmetic(gpu_en, orc_en, means_en, monkeypatch):
    feats, off, marks = _poisoned(means_en)
    assert len(marks) > 60
    _check(gpu_en, orc_en, feats, off, "matrix-core scan")
    monkeypatch.setenv("SSW_SCAN", "fma")
    _check(gpu_en, orc_en, feats, off, "vector-unit scan")
    monkeypatch.delenv("SSW_SCAN")
    monkeypatch.setenv("SSW_MFMA_STEPS", "2")
    _check(gpu_en, orc_en, feats, off, "two steps per wave")
    monkeypatch.delenv("SSW_MFMA_STEPS")
    # a batch large enough for the packed persistent senone kernel (>= 2048 frames)
    big = np.concatenate([feats] * 9)
    boff = np.concatenate([[0]] + [off[1:] + k * off[-1] for k in range(9)]).astype(np.int32)
    got = gpu_en.score_batch(big, boff)
    ref, _, _ = _oracle_batch(orc_en, feats, off)
    for k in range(9):
        assert np.array_equal(got[k * off[-1]:(k + 1) * off[-1]], ref), k


def test_ptm_nonfinite_rows_through_the_exact_chain_kernel(orc_en, means_en, monkeypatch):
    monkeypatch.setenv("SSW_PTM_EXACT", "1")
    g = ssw.Model(ssw.model_dir("en-us"))
    monkeypatch.delenv("SSW_PTM_EXACT")
    feats, off, _ = _poisoned(means_en, seed=7)
    _check(g, orc_en, feats, off, "chain kernel")
    g.close()


def test_ptm_nonfinite_rows_through_the_vtable(gpu_en, orc_en, means_en):
    """vt->frame_eval frame by frame (history carried as the reference carries it)"""
    feats, off, _ = _poisoned(means_en, seed=21)
    feats = feats[:int(off[2])]
    mg = ssw.PtmMgau(gpu_en)
    orc_en.ptm_reset()
    try:
        for t in range(len(feats)):
            mg.frame_idx = t
            orc_en.ptm_set_frame_idx(t)
            got = mg.frame_eval(feats[t], t)
            ref = orc_en.ptm_frame_eval(feats[t], t)
            assert np.array_equal(got, ref), t
    finally:
        mg.free()
        orc_en.ptm_reset()


def test_ms_scorer_infinite_rows(oracle_mod, means_fr, tmp_path):
    import os
    import sys
    from tests.conftest import MODEL_ROOT, ROOT
    from tests.test_cabi_host import synth_mixw_from_sendump
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    src = os.path.join(MODEL_ROOT, "fr-fr")
    orc_fr = oracle_mod.Model(src)
    mixw = str(tmp_path / "mixture_weights")
    synth_mixw_from_sendump(orc_fr, mixw)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
              tmat=os.path.join(src, "transition_matrices"), mixw=mixw)
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    feats = np.concatenate([synth_features(means_fr, 256, 500 + u) for u in range(9)])
    off = (np.arange(10) * 256).astype(np.int32)
    for t, dims, v in ((5, [0], INF), (9, [12], -INF), (40, [3, 20], INF), (300, np.arange(39), INF),
                       (301, [38], -INF), (2000, [13], INF), (2303, [7], np.float32(4.0e19))):
        feats[t, dims] = v
    got = g.score_batch(feats, off, scorer=ssw.SCORER_MS)
    ref = np.concatenate([o.ms_score_utt(feats[off[u]:off[u + 1]]) for u in range(9)])
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert len(bad) == 0, bad[:10].tolist()
    small = g.score_batch(feats[:300], np.array([0, 300], np.int32), scorer=ssw.SCORER_MS)
    assert np.array_equal(small, ref[:300])
    g.close()
