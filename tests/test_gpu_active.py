"""The reference's DEFAULT configuration (compallsen = no) through the batch call
ssw_align_batch_active: per frame only the senones of the search's active HMMs are scored,
through acmod_flags2list's delta list (bridge entries included), codebooks and senones are
normalised over that set, and state_align_search never clears the set.  Checked against the
oracle's per-frame scorer driven by a restatement of state_align_search (as
tests/test_oracle_e2e_goforward.py restates it for the recorded default-configuration scores)."""
import ctypes as C
import os

import numpy as np
import pytest

from soundswallower_amd.synth import synth_alignment_task, synth_features
from tests.conftest import MODEL_ROOT

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def second_pass_active(O, m, feats, senid, tmat, sf, ef, seed_vec, ms=False):
    """state_align_search (src/state_align_search.c:177-268) around the oracle's per-frame scorer
    with compallsen = no; history reset at frame 0.  Returns (rv, states [3n][3], score rows).
    ms: the ms scorer's frame_eval (it writes the listed entries only; acmod's buffer keeps the
    others, which start from 0)."""
    T, n = len(feats), len(senid)
    vec = np.zeros((m.n_sen + 31) // 32, np.uint32) if seed_vec is None else seed_vec.copy()
    W = -(1 << 29)
    sc = np.full((n, 3), W, np.int32)
    hi = np.full((n, 3), -1, np.int32)
    osc = np.full(n, W, np.int32)
    ohi = np.full(n, -1, np.int32)
    best = np.full(n, W, np.int32)
    frame_of = np.full(n, -1, np.int64)
    L = O.lib()
    L.orc_hmm_vit_eval_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8
    L.orc_hmm_vit_eval_many.restype = None
    senid = np.ascontiguousarray(senid, np.uint16)
    tmat = np.ascontiguousarray(tmat, np.int16)
    m.ptm_reset()
    m.ptm_set_frame_idx(0)
    sc[0, 0], hi[0, 0], frame_of[0] = 0, 0, 0
    tokens = np.full((T, n * 3, 2), -1, np.int64)
    rows = np.zeros((T, m.n_sen), np.int16)
    for f in range(T):
        for i in np.nonzero(frame_of == f)[0]:
            for s_ in senid[i]:
                vec[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))
        if ms:
            # the ms scorer writes the LISTED entries only (bridge entries of the delta list
            # included); the batch call returns 0 elsewhere, so does this restatement (acmod's
            # buffer would keep the stale value of a bridge entry that dropped out of the list
            # when a senone between its neighbours joined -- an entry no HMM reads)
            lst = O.flags2list(vec, m.n_sen)
            row = m.ms_frame_eval(feats[f], f, compallsen=False, senone_active=lst)
        else:
            row = m.ptm_frame_eval(feats[f], f, compallsen=False,
                                   senone_active=O.flags2list(vec, m.n_sen))
            m.ptm_set_frame_idx(f + 1)
        rows[f] = row
        idx = np.nonzero(frame_of >= f)[0].astype(np.int32)
        L.orc_hmm_vit_eval_many(m._m, row.ctypes.data, len(idx), idx.ctypes.data, senid.ctypes.data,
                                tmat.ctypes.data, sc.ctypes.data, hi.ctypes.data, osc.ctypes.data,
                                ohi.ctypes.data, best.ctypes.data)
        nf = f + 1
        for i in idx:
            if nf <= ef[i]:
                frame_of[i] = nf
        for i in range(n - 1):
            if frame_of[i] != nf or nf < sf[i + 1]:
                continue
            if frame_of[i + 1] < f or osc[i] > sc[i + 1, 0]:
                sc[i + 1, 0], hi[i + 1, 0], frame_of[i + 1] = osc[i], ohi[i], nf
        for i in np.nonzero(frame_of >= f)[0]:
            for j in range(3):
                tokens[f, i * 3 + j] = (hi[i, j], sc[i, j])
                hi[i, j] = i * 3 + j
    st = np.zeros((n * 3, 3), np.int32)
    last_id, last_sc = int(ohi[n - 1]), int(osc[n - 1])
    if last_id == -1 or frame_of[n - 1] < T:       # "Failed to reach final state"
        return -1, st, rows
    cur_id, last_frame = last_id, T
    for cf in range(T - 2, -1, -1):
        cid, csc = tokens[cf, cur_id]
        if cid == -1:
            return -2, st, rows
        if cid != last_id:
            st[last_id] = (cf + 1, last_frame - (cf + 1), last_sc - csc)
            last_id, last_sc, last_frame = int(cid), int(csc), cf + 1
        cur_id = int(cid)
    st[0, 0], st[0, 1] = 0, last_frame
    return 0, st, rows


def _windows(rng, n_phones, n_frames):
    """word-like windows: groups of phones sharing (sf, ef), non-decreasing, generous overlap"""
    sf = np.zeros(n_phones, np.int32)
    ef = np.full(n_phones, INT_MAX, np.int32)
    i, t = 0, 0
    while i < n_phones:
        g = int(rng.integers(1, 5))
        dur = max(3 * g + 2, int(n_frames * g / n_phones))
        a = max(0, t - 6)
        b = min(n_frames, t + dur + 8)
        sf[i:i + g] = a
        ef[i:i + g] = b
        t += dur
        i += g
    ef = np.maximum.accumulate(ef)
    return sf, ef


def test_align_batch_active_matches_the_restated_search(gpu_en, orc_en, oracle_mod, means_en):
    rng = np.random.default_rng(17)
    cases = [(6, 40, False, False), (9, 70, True, False), (14, 90, True, True), (5, 31, False, True),
             (20, 130, True, True)]
    feats, senids, tmats, sfs, efs, seeds, refs = [], [], [], [], [], [], []
    words = (orc_en.n_sen + 31) // 32
    for k, (n_ph, n_fr, windowed, seeded) in enumerate(cases):
        f = synth_features(means_en, n_fr, 4000 + k)
        senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                              orc_en.n_ciphone, n_ph, 600 + k)
        sf, ef = _windows(rng, n_ph, n_fr) if windowed else (np.zeros(n_ph, np.int32),
                                                             np.full(n_ph, INT_MAX, np.int32))
        seed = np.zeros(words, np.uint32)
        if seeded:                        # what a first pass might have left active
            for s_ in rng.integers(0, orc_en.n_sen, 40):
                seed[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))
        rv, st, rows = second_pass_active(oracle_mod, orc_en, f, senid, tmat, sf, ef,
                                          seed if seeded else None)
        feats.append(f); senids.append(senid); tmats.append(tmat); sfs.append(sf); efs.append(ef)
        seeds.append(seed); refs.append((rv, st, rows))
    frame_off = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum([len(s) for s in senids])]).astype(np.int32)
    allf = np.concatenate(feats)
    d_feats = gpu_en.to_device(allf)
    d_scr = gpu_en.device_malloc(len(allf) * gpu_en.n_sen * 2)
    try:
        st, status = gpu_en.align_batch_active(d_feats, frame_off, phone_off, np.concatenate(senids),
                                               np.concatenate(tmats), np.concatenate(sfs),
                                               np.concatenate(efs), seed_active=np.stack(seeds),
                                               d_senscr=d_scr)
        scr = np.zeros((len(allf), gpu_en.n_sen), np.int16)
        gpu_en._L.ssw_memcpy_d2h(scr.ctypes.data, d_scr, scr.nbytes)
    finally:
        gpu_en.device_free(d_feats)
        gpu_en.device_free(d_scr)
    assert any(r[0] == 0 for r in refs)
    for u, (rv, rst, rows) in enumerate(refs):
        a, b = frame_off[u], frame_off[u + 1]
        bad = np.argwhere(scr[a:b] != rows)
        assert len(bad) == 0, ("scores", u, len(bad), bad[:6].tolist(),
                               [(int(scr[a + i, j]), int(rows[i, j])) for i, j in bad[:6]])
        assert (status[u] == 0) == (rv == 0), u
        if rv == 0:
            assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), ("states", u)


def test_default_configuration_scores_from_the_batch_call(gpu_en, orc_en, oracle_mod):
    """the phone scores the real library printed for goforward in its default
    configuration.  First pass (and with it the seed set it leaves active) from the oracle's
    restatement, second pass = ONE ssw_align_batch_active call."""
    from oracle import fsg_oracle as F
    from tests.test_oracle_e2e_goforward import (REF_SCORES_DEFAULT, REF_WORDS, goforward_features,
                                                 populate)
    O, m = oracle_mod, orc_en
    feats = goforward_features(O)
    d = os.path.join(MODEL_ROOT, "en-us")
    lex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    vec = np.zeros((m.n_sen + 31) // 32, np.uint32)
    m.ptm_reset()
    m.ptm_set_frame_idx(0)

    def first_pass_scores(f, sen):
        vec[:] = 0
        for s_ in np.unique(np.asarray(sen).ravel()):
            vec[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))
        row = m.ptm_frame_eval(feats[f], f, compallsen=False, senone_active=O.flags2list(vec, m.n_sen))
        m.ptm_set_frame_idx(f + 1)
        return row

    seg = F.first_pass(m, lex, "go forward ten meters".split(), first_pass_scores, n_frames=len(feats))
    assert [(w, s, e - s + 1) for (w, s, e, _) in seg] == [(w, s, dd) for (w, s, dd, _) in REF_WORDS]
    words = [(w, s, e - s + 1) for (w, s, e, _) in seg]
    phones = populate(O, m, words)
    senid = np.ascontiguousarray(m.sseq[[p[1] for p in phones]], np.uint16)
    tmat = np.array([p[2] for p in phones], np.int16)
    wstart = np.array([words[p[3]][1] for p in phones])
    wdur = np.array([words[p[3]][2] for p in phones])
    sf = np.where(wstart > 0, wstart, 0).astype(np.int32)
    ef = np.where(wdur > 0, wstart + wdur, INT_MAX).astype(np.int32)
    d_feats = gpu_en.to_device(feats)
    try:
        st, status = gpu_en.align_batch_active(d_feats, [0, len(feats)], [0, len(phones)], senid, tmat,
                                               sf, ef, seed_active=vec[None])
    finally:
        gpu_en.device_free(d_feats)
    assert status[0] == 0
    ph_score = st.reshape(len(phones), 3, 3)[:, :, 2].sum(1)
    assert [int(x) for x in ph_score] == REF_SCORES_DEFAULT


def test_ms_scorer_through_the_batched_active_set_path(oracle_mod, orc_fr, means_fr, tmp_path):
    """VERDICT r2 "missing" 3: ssw_align_batch_active_ex with SSW_SCORER_MS -- the active-list
    half of ms_cont_mgau_frame_eval (src/ms_mgau.c:322-365) for a whole batch: listed senones
    only, best of them subtracted with the clamp, never-listed entries 0.  Against the restated
    search around the oracle's ms scorer: every score row's LISTED entries and every state entry."""
    import soundswallower_amd as ssw
    from tests.test_cabi_host import synth_mixw_from_sendump
    src = os.path.join(MODEL_ROOT, "fr-fr")
    mixw = str(tmp_path / "mixture_weights")
    synth_mixw_from_sendump(orc_fr, mixw)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
              tmat=os.path.join(src, "transition_matrices"), mixw=mixw)
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    rng = np.random.default_rng(23)
    cases = [(7, 50, False), (12, 90, True), (20, 150, True)]
    feats, senids, tmats, sfs, efs, refs = [], [], [], [], [], []
    for k, (n_ph, n_fr, windowed) in enumerate(cases):
        f = synth_features(means_fr, n_fr, 5100 + k)
        senid, tmat, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone,
                                              n_ph, 900 + k)
        sf, ef = _windows(rng, n_ph, n_fr) if windowed else (np.zeros(n_ph, np.int32),
                                                             np.full(n_ph, INT_MAX, np.int32))
        refs.append(second_pass_active(oracle_mod, o, f, senid, tmat, sf, ef, None, ms=True))
        feats.append(f); senids.append(senid); tmats.append(tmat); sfs.append(sf); efs.append(ef)
    frame_off = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int32)
    phone_off = np.concatenate([[0], np.cumsum([len(s) for s in senids])]).astype(np.int32)
    allf = np.concatenate(feats)
    d_feats = g.to_device(allf)
    d_scr = g.device_malloc(len(allf) * g.n_sen * 2)
    try:
        st, status = g.align_batch_active(d_feats, frame_off, phone_off, np.concatenate(senids),
                                          np.concatenate(tmats), np.concatenate(sfs),
                                          np.concatenate(efs), d_senscr=d_scr, scorer=ssw.SCORER_MS)
        scr = np.zeros((len(allf), g.n_sen), np.int16)
        g._L.ssw_memcpy_d2h(scr.ctypes.data, d_scr, scr.nbytes)
    finally:
        g.device_free(d_feats)
        g.device_free(d_scr)
    assert any(r[0] == 0 for r in refs)
    for u, (rv, rst, rows) in enumerate(refs):
        a, b = frame_off[u], frame_off[u + 1]
        bad = np.argwhere(scr[a:b] != rows)
        assert len(bad) == 0, ("scores", u, len(bad), bad[:6].tolist(),
                               [(int(scr[a + i, j]), int(rows[i, j])) for i, j in bad[:6]])
        assert (status[u] == 0) == (rv == 0), u
        if rv == 0:
            assert np.array_equal(st[phone_off[u] * 3:phone_off[u + 1] * 3], rst), ("states", u)
