"""GPU parity: the ms scorer (gauden_dist + senone_eval) vs the CPU oracle, fr-fr with a
mixture_weights file synthesised from the sendump (BASELINE config 4)."""
import os

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import synth_features
from tests.conftest import MODEL_ROOT
from tests.test_cabi_host import synth_mixw_from_sendump

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ms_models(oracle_mod, orc_fr, tmp_path_factory):
    src = os.path.join(MODEL_ROOT, "fr-fr")
    mixw = str(tmp_path_factory.mktemp("ms") / "mixture_weights")
    synth_mixw_from_sendump(orc_fr, mixw)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
              tmat=os.path.join(src, "transition_matrices"), mixw=mixw)
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    return g, o


def test_ms_small_batch_bit_exact(ms_models, means_fr):
    g, o = ms_models
    lens = [33, 1, 70]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = np.concatenate([synth_features(means_fr, n, 12345 + i) for i, n in enumerate(lens)])
    got = g.score_batch(feats, off, scorer=ssw.SCORER_MS)
    ref = o.ms_score_utt(feats)
    assert np.array_equal(got, ref)


def test_ms_config4_8192_frames(ms_models, means_fr):
    """BASELINE config 4: 8192 frames as 32 x 256."""
    g, o = ms_models
    feats = np.concatenate([synth_features(means_fr, 256, 12345 + i) for i in range(32)])
    off = (np.arange(33) * 256).astype(np.int32)
    got = g.score_batch(feats, off, scorer=ssw.SCORER_MS)
    flagged, pairs = g.last_stats()
    assert pairs == 8192 * 108
    # every one of the 8192 rows against the committed oracle checksums (tests/golden/
    # make_golden.py:config4), and the first 1024 against the oracle run here
    import json
    import zlib
    from tests.conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "synthetic_oracle.json")) as fh:
        gold = json.load(fh)["config4_fr_fr_ms"]
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
    assert crc(feats) == gold["feats_crc"]
    assert [crc(got[u * 256:(u + 1) * 256]) for u in range(32)] == gold["utt_crc"]
    assert crc(got) == gold["crc"]
    ref = o.ms_score_utt(feats[:1024])
    assert np.array_equal(got[:1024], ref)
    assert (got.min(axis=1) == 0).all()


@pytest.mark.parametrize("knobs", [{"SSW_SEN_GENERIC": "1"}, {"SSW_SCAN": "fma"}])
def test_ms_tuning_knobs_do_not_change_results(ms_models, means_fr, monkeypatch, knobs):
    """The run-time-stream-count instance of ms_senone_kernel and the vector-unit scan give
    config 4's committed checksums as well."""
    import json
    import zlib
    from tests.conftest import ROOT
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g, _ = ms_models
    with open(os.path.join(ROOT, "tests", "golden", "synthetic_oracle.json")) as fh:
        gold = json.load(fh)["config4_fr_fr_ms"]
    feats = np.concatenate([synth_features(means_fr, 256, 12345 + i) for i in range(32)])
    off = (np.arange(33) * 256).astype(np.int32)
    got = g.score_batch(feats, off, scorer=ssw.SCORER_MS)
    assert zlib.crc32(np.ascontiguousarray(got).tobytes()) & 0xFFFFFFFF == gold["crc"]


def test_ms_exact_ties_take_the_exact_pass(ms_models, means_fr):
    """A frame sitting exactly on two identical densities is impossible to build from the model,
    but duplicated frames and extreme features must still agree with the oracle."""
    g, o = ms_models
    base = synth_features(means_fr, 4, 9)
    feats = np.concatenate([np.repeat(base[:1], 5, 0), base * 50.0, -base * 1000.0])
    got = g.score_batch(feats, scorer=ssw.SCORER_MS)
    assert np.array_equal(got, o.ms_score_utt(feats))


def test_ms_mgau_vtable(ms_models, means_fr):
    g, o = ms_models
    mg = ssw.MsMgau(g)
    assert mg.name == "ms"
    feats = synth_features(means_fr, 3, 1)
    for t in range(3):
        assert np.array_equal(mg.frame_eval(feats[t], t), o.ms_frame_eval(feats[t], t))
    mg.free()


def test_ptm_model_without_mixw_refuses_ms(gpu_en):
    with pytest.raises(ssw.SswError, match="mixture_weights"):
        gpu_en.score_batch(np.zeros((2, 39), np.float32), scorer=ssw.SCORER_MS)
    with pytest.raises(ssw.SswError):
        ssw.MsMgau(gpu_en)


def test_ms_mgau_vtable_compallsen_no(ms_models, means_fr, oracle_mod):
    g, o = ms_models
    mg = ssw.MsMgau(g)
    rng = np.random.default_rng(9)
    feats = synth_features(means_fr, 4, 2)
    for t, dens in enumerate((0.01, 0.2, 1.0, 0.05)):
        vec = np.zeros((o.n_sen + 31) // 32, np.uint32)
        for s in np.flatnonzero(rng.random(o.n_sen) < dens):
            vec[s // 32] |= np.uint32(1 << (s % 32))
        lst = oracle_mod.flags2list(vec, o.n_sen)
        got = mg.frame_eval(feats[t], t, compallsen=False, senone_active=lst)
        ref = o.ms_frame_eval(feats[t], t, compallsen=False, senone_active=lst)
        assert np.array_equal(got, ref), t
    mg.free()


def test_ms_packed_senone_kernel_and_its_marked_frames(ms_models, means_fr, monkeypatch):
    """Round 3: batches of the ms scorer go through the packed persistent senone kernel (the PTM
    kernel's machinery, 16 bits per slot).  What 16 bits cannot hold -- a density at or below
    logmath's zero, where logmath_add's short-cuts apply (src/logmath.c:228-272), or sums beyond
    int16, where src/ms_mgau.c:314-320 clamps -- marks the frame, and marked frames are redone
    with the reference's own arithmetic.  Features scaled by 100 .. 3000 put densities below
    -5.4e8 (zero << 10) and below -2^31 (the clamp of src/ms_senone.c:332-335); they are mixed
    with ordinary frames so that marked and unmarked frames share units.  Against the oracle and
    against ms_senone_kernel (SSW_MS_SENONE=old), frame by frame."""
    g, o = ms_models
    rng = np.random.default_rng(5)
    base = synth_features(means_fr, 2560, 321)
    scale = np.ones(len(base), np.float32)
    pick = rng.random(len(base)) < 0.08
    scale[pick] = rng.choice(np.array([100.0, 300.0, 1000.0, 3000.0, 30.0], np.float32), int(pick.sum()))
    feats = np.ascontiguousarray(base * scale[:, None], np.float32)
    feats[7] = 0.0
    got = g.score_batch(feats, scorer=ssw.SCORER_MS)
    monkeypatch.setenv("SSW_MS_SENONE", "old")
    old = g.score_batch(feats, scorer=ssw.SCORER_MS)
    monkeypatch.delenv("SSW_MS_SENONE")
    bad = np.nonzero((got != old).any(axis=1))[0]
    assert len(bad) == 0, (bad[:10], scale[bad[:10]])
    ref = o.ms_score_utt(feats[:400])
    assert np.array_equal(got[:400], ref)
    sel = np.nonzero(pick)[0][:40]
    for t in sel:                       # the extreme frames one by one against the oracle
        assert np.array_equal(got[t], o.ms_score_utt(feats[t:t + 1])[0]), (t, scale[t])
