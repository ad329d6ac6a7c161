"""fast_logmath_add with a running value below zero (tied_mgau_common.h:100-117).

`x <- min(x, y) - table[|x - y|]` starts from `mixw + score`, both >= 0, and table[0] = 7 for the
8-bit table: two equal entries below 7 leave x negative, after which |x - y| can pass 255, where
the reference reads past its table (the oracle defines that read as 0, the value every entry from
29 up holds).  The shipped sendumps have almost no weights that small, so this case needs a
sendump of its own: an 8-bit dump with a third of its weights in 0..3.  All three PTM senone code
paths are driven: the batch kernel (biased 16-bit chain), the one-frame vtable kernel with and
without an active list, and the batched active-set kernel (the last two used a 16-bit |x - y|
until round 3)."""
import os
import struct

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import synth_alignment_task, synth_features
from tests.conftest import MODEL_ROOT

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def write_sendump8(path, mixw, n_feat, n_density, n_sen):
    """A plain 8-bit sendump (src/ptm_mgau.c:456-609): title, header, key strings, a zero length,
    rows, columns, then [feat][density][n_sen] bytes."""
    def s(txt):
        b = txt.encode() + b"\0"
        return struct.pack("<i", len(b)) + b
    blob = s("s3 senone dump") + s("synthetic, tiny weights")
    for kv in (f"feature_count {n_feat}", f"mixture_count {n_density}", f"model_count {n_sen}"):
        blob += s(kv)
    blob += struct.pack("<i", 0) + struct.pack("<ii", n_density, n_sen)
    blob += np.ascontiguousarray(mixw, np.uint8).tobytes()
    with open(path, "wb") as fh:
        fh.write(blob)


@pytest.fixture(scope="module")
def tiny_weight_models(oracle_mod, orc_en, tmp_path_factory):
    src = os.path.join(MODEL_ROOT, "en-us")
    sd = str(tmp_path_factory.mktemp("sd") / "sendump8")
    rng = np.random.default_rng(11)
    shape = (orc_en.n_feat, orc_en.n_density, orc_en.n_sen)
    mixw = rng.integers(0, 256, shape, dtype=np.uint8)
    small = rng.random(shape) < 0.35
    mixw[small] = rng.integers(0, 4, int(small.sum()), dtype=np.uint8)
    write_sendump8(sd, mixw, *shape)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"), sendump=sd,
              tmat=os.path.join(src, "transition_matrices"))
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    assert np.array_equal(g.table("ptm_mixw").reshape(shape), mixw)
    return g, o


def _near_tie_features(means, n, seed):
    """frames between two densities of a codebook: the two best scores then share their
    `>> 10` bucket, so two codewords enter the chain with the same normalised score"""
    rng = np.random.default_rng(seed)
    base = synth_features(means, n, seed)
    for i in range(n):
        cb = int(rng.integers(0, means.shape[0]))
        for f in range(3):
            d1, d2 = rng.choice(means.shape[2], 2, replace=False)
            base[i, f * 13:(f + 1) * 13] = (means[cb, f, d1] + means[cb, f, d2]) * 0.5
    return np.ascontiguousarray(base, np.float32)


def test_the_case_occurs(tiny_weight_models, means_en):
    """the oracle's own chain does go below zero on these inputs (otherwise the tests below
    would not test what they claim): recompute one frame's chains in numpy"""
    g, o = tiny_weight_models
    feats = _near_tie_features(means_en, 24, 5)
    out, cw, sc = o.ptm_score_utt(feats, want_topn=True)
    mixw = g.table("ptm_mixw").reshape(o.n_feat, o.n_density, o.n_sen).astype(np.int64)
    tab = o.logadd_table_8b.astype(np.int64)
    tab = np.concatenate([tab, np.zeros(1024, np.int64)])
    sen2cb = o.sen2cimap
    below = 0
    for t in range(len(feats)):
        q = sc[t].reshape(-1, o.n_feat, 4)      # already normalised (codebook_norm works in place)
        c = cw[t].reshape(-1, o.n_feat, 4)
        tot = np.zeros(o.n_sen, np.int64)
        for f in range(o.n_feat):
            x = mixw[f, c[sen2cb, f, 0], np.arange(o.n_sen)] + q[sen2cb, f, 0]
            for k in range(1, 4):
                y = mixw[f, c[sen2cb, f, k], np.arange(o.n_sen)] + q[sen2cb, f, k]
                x = np.minimum(x, y) - tab[np.abs(x - y)]
                below += int((x < 0).sum())
            tot += x
        assert np.array_equal((tot - tot.min()).astype(np.int16), out[t])   # the restatement holds
    assert below > 100, below


def test_batch_kernel(tiny_weight_models, means_en):
    g, o = tiny_weight_models
    feats = np.concatenate([_near_tie_features(means_en, 150, 5), synth_features(means_en, 150, 6)])
    assert np.array_equal(g.score_batch(feats), o.ptm_score_utt(feats))


def test_one_frame_vtable_kernel(tiny_weight_models, oracle_mod, means_en):
    g, o = tiny_weight_models
    feats = _near_tie_features(means_en, 12, 7)
    for fused in ("0", "1"):
        os.environ["SSW_FRAME_FUSED"] = fused
        try:
            mg = ssw.PtmMgau(g)
        finally:
            del os.environ["SSW_FRAME_FUSED"]
        o.ptm_reset()
        rng = np.random.default_rng(3)
        for t in range(len(feats)):
            mg.frame_idx = t
            o.ptm_set_frame_idx(t)
            if t % 2 == 0:
                got = mg.frame_eval(feats[t], t)
                ref = o.ptm_frame_eval(feats[t], t)
            else:
                vec = np.zeros((o.n_sen + 31) // 32, np.uint32)
                for s_ in rng.integers(0, o.n_sen, 700):
                    vec[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))
                lst = oracle_mod.flags2list(vec, o.n_sen)
                got = mg.frame_eval(feats[t], t, compallsen=False, senone_active=lst)
                ref = o.ptm_frame_eval(feats[t], t, compallsen=False, senone_active=lst)
            assert np.array_equal(got, ref), (fused, t)
        mg.free()


def test_batched_active_set_kernel(tiny_weight_models, oracle_mod, means_en):
    from tests.test_gpu_active import second_pass_active
    g, o = tiny_weight_models
    n_ph, n_fr = 12, 80
    feats = _near_tie_features(means_en, n_fr, 9)
    senid, tmat, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone, n_ph, 41)
    sf, ef = np.zeros(n_ph, np.int32), np.full(n_ph, INT_MAX, np.int32)
    rv, rst, rows = second_pass_active(oracle_mod, o, feats, senid, tmat, sf, ef, None)
    d_feats = g.to_device(feats)
    d_scr = g.device_malloc(len(feats) * g.n_sen * 2)
    try:
        st, status = g.align_batch_active(d_feats, [0, n_fr], [0, n_ph], senid, tmat, sf, ef,
                                          d_senscr=d_scr)
        scr = np.zeros((n_fr, g.n_sen), np.int16)
        g._L.ssw_memcpy_d2h(scr.ctypes.data, d_scr, scr.nbytes)
    finally:
        g.device_free(d_feats)
        g.device_free(d_scr)
    assert np.array_equal(scr, rows)
    assert (status[0] == 0) == (rv == 0)
    if rv == 0:
        assert np.array_equal(st, rst)
