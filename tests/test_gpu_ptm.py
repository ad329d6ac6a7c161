"""GPU parity: PTM senone scoring through the C ABI vs the CPU oracle (bit-exact int16)."""
import numpy as np
import pytest

from soundswallower_amd.synth import synth_features

pytestmark = pytest.mark.gpu


def _oracle_batch(orc, feats, utt_off):
    outs, cws, scs = [], [], []
    for u in range(len(utt_off) - 1):
        o, cw, sc = orc.ptm_score_utt(feats[utt_off[u]:utt_off[u + 1]], want_topn=True)
        outs.append(o)
        cws.append(cw)
        scs.append(sc)
    return np.concatenate(outs), np.concatenate(cws), np.concatenate(scs)


def test_tables_match_oracle(gpu_en, orc_en):
    for name, ref in (("mean", orc_en.mean), ("var", orc_en.var), ("det", orc_en.det.reshape(-1)),
                      ("ptm_mixw", orc_en.ptm_mixw.reshape(-1)), ("tp", orc_en.tp.reshape(-1)),
                      ("sseq", orc_en.sseq.reshape(-1)), ("sen2cb", orc_en.sen2cimap),
                      ("logadd8", orc_en.logadd_table_8b.astype(np.uint8))):
        got = gpu_en.table(name)
        assert got.shape == ref.shape, name
        assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(ref).view(np.uint8)), name


def test_ptm_small_batch_bit_exact(gpu_en, orc_en, means_en):
    lens = [40, 1, 17, 64]
    utt_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = np.concatenate([synth_features(means_en, n, 12345 + i) for i, n in enumerate(lens)])
    got = gpu_en.score_batch(feats, utt_off)
    ref, rcw, rsc = _oracle_batch(orc_en, feats, utt_off)
    gcw, gsc = gpu_en.last_topn(len(feats))
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert got.dtype == np.int16 and got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_ptm_config2_layouts_bit_exact(gpu_en, orc_en, means_en):
    """BASELINE config 2: 4096 frames as 16 x 256 and as 1 x 4096, full int16 parity, and the
    top-N state (codeword order) of every frame."""
    feats = np.concatenate([synth_features(means_en, 256, 12345 + i) for i in range(16)])
    for utt_off in (np.arange(17, dtype=np.int32) * 256, np.array([0, 4096], np.int32)):
        got = gpu_en.score_batch(feats, utt_off)
        flagged, pairs = gpu_en.last_stats()
        gcw, _ = gpu_en.last_topn(len(feats))
        ref, rcw, _ = _oracle_batch(orc_en, feats, utt_off)
        assert pairs == 4096 * 126
        assert 0 < flagged < pairs // 20, "exact pass should be rare but not empty here"
        assert np.array_equal(gcw.astype(np.int32), rcw)
        assert np.array_equal(got, ref)


def test_ptm_correlated_frames_bit_exact(gpu_en, orc_en, means_en):
    """Slowly drifting features (speech-like): carried top-N codewords stay good, the regime in
    which the reference's history matters most."""
    base = synth_features(means_en, 8, 777)
    t = np.linspace(0, 1, 300, dtype=np.float32)[:, None]
    feats = (base[0] * (1 - t) + base[1] * t).astype(np.float32)
    feats = np.concatenate([feats, np.repeat(base[2:3], 40, axis=0)])  # identical frames too
    got = gpu_en.score_batch(feats)
    ref = orc_en.ptm_score_utt(feats)
    assert np.array_equal(got, ref)
