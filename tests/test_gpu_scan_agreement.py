"""The two speculative scans against each other.  The vector-unit scan's
error bound is replayed bit for bit on the CPU (tests/test_scan_bound.py); the matrix-core scan's
rests on an assumed accumulation error of v_mfma_f32_32x32x16_f16 (csrc/ssw_model.c).  A key
below the true density would drop a top-4 candidate -- the one failure the exact pass cannot
catch -- and the two scans would then disagree.  So: the same batches through SSW_SCAN=fma and
through the default (matrix cores), top-N codewords, raw scores and senone rows compared entry by
entry, on BASELINE config 2's frames, on the stress batch of tests/test_gpu_ptm.py, on inputs
built to cancel in the matrix-core accumulation, and (ms scorer) on fr-fr."""
import numpy as np
import pytest

from soundswallower_amd.synth import synth_features

pytestmark = pytest.mark.gpu


def _both(gpu, feats, monkeypatch, scorer=None):
    kw = {} if scorer is None else {"scorer": scorer}
    monkeypatch.setenv("SSW_SCAN", "fma")
    a = gpu.score_batch(feats, **kw)
    acw, asc = gpu.last_topn(len(feats))
    a_stats = gpu.last_stats()
    monkeypatch.delenv("SSW_SCAN")
    b = gpu.score_batch(feats, **kw)
    bcw, bsc = gpu.last_topn(len(feats))
    b_stats = gpu.last_stats()
    if not np.array_equal(acw, bcw):
        bad = np.argwhere((acw != bcw).any(axis=3))
        t, cb, f = (int(v) for v in bad[0])
        x = np.asarray(feats, np.float32).reshape(len(acw), -1)[t, 13 * f:13 * f + 13]
        raise AssertionError("top-N codewords differ between the scans: %d (frame, codebook, stream) "
                             "entries, first at frame %d codebook %d stream %d (max |x| %.4g): %s / %s"
                             % (len(bad), t, cb, f, float(np.abs(x).max()), acw[t, cb, f], bcw[t, cb, f]))
    assert np.array_equal(asc, bsc), "top-N scores differ between the scans"
    assert np.array_equal(a, b)
    return a_stats, b_stats


def _stress(means, rec_var):
    rng = np.random.default_rng(2024)
    base = synth_features(means, 64, 4321)
    parts = [base * 8.0, base * 40.0, base * 0.01, -base, np.zeros((4, 39), np.float32),
             rng.normal(0, 3, (64, 39)).astype(np.float32),
             rng.normal(0, 30, (32, 39)).astype(np.float32)]
    # exactly on twice the mean of a density: every a x + b x^2 pair of the key cancels
    n_cb, n_feat, n_den, _ = means.shape
    two = np.empty((256, 39), np.float32)
    for i in range(len(two)):
        cb, d = int(rng.integers(0, n_cb)), int(rng.integers(0, n_den))
        for f in range(n_feat):
            two[i, f * 13:(f + 1) * 13] = 2 * means[cb, f, d]
    parts += [two, two * np.float32(1.01)]
    return np.ascontiguousarray(np.concatenate(parts), np.float32)


def test_config2_frames(gpu_en, means_en, monkeypatch):
    feats = np.concatenate([synth_features(means_en, 256, 12345 + i) for i in range(16)])
    (fa, pa), (fb, pb) = _both(gpu_en, feats, monkeypatch)
    assert pa == pb == 4096 * 126
    assert 0 < fa < pa // 100 and 0 < fb < pb // 100      # the exact pass stays rare in both


def test_stress_and_cancelling_inputs(gpu_en, means_en, monkeypatch):
    feats = _stress(means_en, None)
    reps = -(-2304 // len(feats))     # enough frames for the matrix-core scan to be chosen
    feats = np.concatenate([feats] * reps)
    _both(gpu_en, feats, monkeypatch)


def test_ms_scorer_fr_fr(gpu_fr, orc_fr, means_fr, monkeypatch, tmp_path):
    import os
    import soundswallower_amd as ssw
    from tests.conftest import MODEL_ROOT
    from tests.test_cabi_host import synth_mixw_from_sendump
    src = os.path.join(MODEL_ROOT, "fr-fr")
    mixw = str(tmp_path / "mixture_weights")
    synth_mixw_from_sendump(orc_fr, mixw)
    g = ssw.Model(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
                  variances=os.path.join(src, "variances"), mixw=mixw,
                  tmat=os.path.join(src, "transition_matrices"))
    feats = np.concatenate([synth_features(means_fr, 2304, 77), _stress(means_fr, None)])
    _both(g, feats, monkeypatch, scorer=ssw.SCORER_MS)
    _both(gpu_fr, feats, monkeypatch)          # and the PTM scorer on the same model
