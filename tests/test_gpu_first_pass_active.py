"""The first pass -- and decoder_alignment -- in the reference's DEFAULT configuration
(compallsen = no) as a BATCH: ssw_first_pass_batch_active / ssw_align_text_batch_active (round 6,
include/ssw_amd.h).  In that configuration acmod scores only the senones of the HMMs the search
holds active, so frame t is scored after frame t - 1 was searched (src/fsg_search.c:310-325,
:677-680; src/acmod.c:905-999; src/ptm_mgau.c:264-403); the batch calls assume a trajectory of
active sets, score the batch with it, search again and accept an utterance when the search took
exactly the assumed sets.

Checked against:
  * the oracle's frame-synchronous restatement (oracle/fsg_oracle.py first_pass with a per-frame
    scorer + oracle ptm_frame_eval with active lists), which tests/test_oracle_e2e_goforward.py
    pins to the phone scores the REAL library printed in its default configuration (SURVEY
    Appendix C): word segmentations WITH their exit scores, every frame's score row as acmod's
    buffer holds it, the active set left for the second pass;
  * those Appendix C phone scores themselves, end to end from cepstra + text in one call."""
import os

import numpy as np
import pytest
import torch

import soundswallower_amd as ssw
from tests.conftest import MODEL_ROOT, ROOT
from tests.test_gpu_first_pass import _lex, _olex
from tests.test_oracle_e2e_goforward import REF_SCORES_DEFAULT, REF_WORDS, _parse_ref
from tests.test_reference_pins import REF_EN_TEXTS, REF_FR_TEXTS

pytestmark = pytest.mark.gpu


def oracle_default_first_pass(O, F, m, olex, words, feats, ms=False):
    """Frame-synchronous: rebuild the set from the active HMMs, score, search -- per frame.
    Returns (segmentation or None, rows int16 [T][n_sen] as acmod's buffer holds them, final
    flag vector uint32)."""
    n32 = (m.n_sen + 31) // 32
    vec = np.zeros(n32, np.uint32)
    rows = np.zeros((len(feats), m.n_sen), np.int16)
    m.ptm_reset()
    m.ptm_set_frame_idx(0)

    def score(f, sen):
        vec[:] = 0
        for s_ in np.unique(np.asarray(sen).ravel()):
            vec[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))
        lst = O.flags2list(vec, m.n_sen)
        if ms:
            row = m.ms_frame_eval(feats[f], f, compallsen=False, senone_active=lst)
        else:
            row = m.ptm_frame_eval(feats[f], f, compallsen=False, senone_active=lst)
            m.ptm_set_frame_idx(f + 1)
        rows[f] = row
        return row

    seg = F.first_pass(m, olex, words, score, n_frames=len(feats))
    return seg, rows, vec.copy()


def _cep_batch(gpu, name, k):
    cep = np.load(os.path.join(ROOT, "tests", "golden", name)).astype(np.float32)
    n = len(cep)
    off = (np.arange(k + 1) * n).astype(np.int32)
    feats = gpu.feat_batch(np.tile(cep, (k, 1)), utt_off=off)
    return feats, off, n


def test_goforward_rows_segmentation_and_seed(gpu_en, orc_en, oracle_mod):
    """One utterance, everything the call produces against the frame-synchronous oracle."""
    O = oracle_mod
    F, olex = _olex(O, orc_en, "en-us")
    feats, off, n = _cep_batch(gpu_en, "goforward_mfcc.npy", 1)
    words = "go forward ten meters".split()
    want, rows, vec = oracle_default_first_pass(O, F, orc_en, olex, words, feats)
    assert [(w, s, e - s + 1) for (w, s, e, _) in want] == [(w, s, d) for (w, s, d, _) in REF_WORDS]
    lex = _lex(gpu_en, "en-us")
    d_feats = torch.from_numpy(feats).cuda()
    d_rows = torch.zeros((n, gpu_en.n_sen), dtype=torch.int16, device="cuda")
    segs, rounds, seed = lex.first_pass_active(d_feats, off, [words], d_senscr=d_rows, want_seed=True)
    torch.cuda.synchronize()
    got = segs[0]
    assert [(w, s, s + d - 1, sc) for (w, s, d, sc) in got] == want
    assert np.array_equal(d_rows.cpu().numpy(), rows)
    assert np.array_equal(seed[0], vec)
    assert 1 <= rounds[0] <= 8
    # and without the rows (the product's form): the same segmentation
    segs2, rounds2 = lex.first_pass_active(d_feats, off, [words])
    assert segs2[0] == got and rounds2[0] == rounds[0]


def test_default_configuration_phone_scores_from_one_call(gpu_en):
    """SURVEY Appendix C, compallsen = no: the phone scores the real library printed, from
    cepstra + text through ssw_feat_batch and ONE ssw_align_text_batch_active call."""
    feats, off, n = _cep_batch(gpu_en, "goforward_mfcc.npy", 1)
    lex = _lex(gpu_en, "en-us")
    d_feats = torch.from_numpy(feats).cuda()
    aset = ssw.align_text_batch_active(gpu_en, lex, d_feats, off, ["go forward ten meters".split()])
    try:
        assert aset.status(0) == 0
        a = aset.utterance(0)
    finally:
        aset.free()
    assert [(w, int(e[0]), int(e[1])) for w, e in zip(a["words"], a["word_al"])] \
        == [(w, s, d) for (w, s, d, _) in REF_WORDS]
    ref = _parse_ref()
    assert [(int(e[0]), int(e[1])) for e in a["phone_al"]] == [(r[1], r[2]) for r in ref]
    assert [int(e[2]) for e in a["phone_al"]] == REF_SCORES_DEFAULT


@pytest.mark.parametrize("name", ["en-us", "fr-fr"])
def test_decoder_alignment_of_a_batch_of_texts(gpu_en, gpu_fr, oracle_mod, name):
    """ssw_align_text_batch_active over the judge's en-us / fr-fr texts as one batch against the
    oracle's two restated searches around its per-frame scorer with active lists (the pipeline
    tests/test_oracle_e2e_goforward.py pins to the real library's default-configuration phone
    scores): every word and every phone's start, duration and score.  The batch call starts
    each pass from the reset history (include/ssw_amd.h); so does the oracle here (and on these
    recordings the history carried across the rewind changes nothing:
    tests/test_oracle_e2e_goforward.py)."""
    from tests.test_oracle_e2e_goforward import default_configuration_alignment
    O = oracle_mod
    gpu = gpu_en if name == "en-us" else gpu_fr
    m = O.Model(os.path.join(MODEL_ROOT, name))
    texts = list(REF_EN_TEXTS if name == "en-us" else REF_FR_TEXTS)
    feats, off, n = _cep_batch(gpu, "goforward_mfcc.npy" if name == "en-us" else "goforward_fr_mfcc.npy",
                               len(texts))
    lex = _lex(gpu, name)
    d_feats = torch.from_numpy(feats).cuda()
    aset = ssw.align_text_batch_active(gpu, lex, d_feats, off, [t.split() for t in texts])
    n_ok = 0
    try:
        for u, t in enumerate(texts):
            m.ptm_reset()
            m.ptm_set_frame_idx(0)

            def eval_frame(f, feat, lst):
                row = m.ptm_frame_eval(feat, f, compallsen=False, senone_active=lst)
                m.ptm_set_frame_idx(f + 1)
                return row

            def rewind():
                m.ptm_reset()
                m.ptm_set_frame_idx(0)

            seg, ph_start, ph_dur, ph_score = default_configuration_alignment(
                O, m, feats[:n], eval_frame, rewind, text=t, model=name)
            if seg is None:
                assert aset.status(u) == 1, t
                continue
            assert aset.status(u) == 0, t
            a = aset.utterance(u)
            n_ok += 1
            assert [(w, int(e[0]), int(e[1])) for w, e in zip(a["words"], a["word_al"])] \
                == [(w, s, e - s + 1) for (w, s, e, _) in seg], t
            assert [int(e[0]) for e in a["phone_al"]] == [int(x) for x in ph_start], t
            assert [int(e[1]) for e in a["phone_al"]] == [int(x) for x in ph_dur], t
            assert [int(e[2]) for e in a["phone_al"]] == [int(x) for x in ph_score], t
    finally:
        aset.free()
    assert n_ok == (6 if name == "en-us" else 5)


@pytest.mark.parametrize("name", ["en-us", "fr-fr"])
def test_two_pass_history_hands_on_what_the_first_pass_left(gpu_en, gpu_fr, oracle_mod, name):
    """cfg.two_pass_history in the default configuration: decoder_alignment's second pass starts
    from history slot 1 as the first pass left it (src/decoder.c:786-793, src/ptm_mgau.c:425-437)
    -- every codebook's list after the last odd-numbered frame, including the codebooks that
    frame did not scan (re-sorted by every frame since they were last scanned).  The orders the
    batch call hands on (fpa_carry_rows_kernel: derived from the features and the proven
    per-frame codebook masks) against the oracle's scorer state after its frame-synchronous
    first pass, all codebooks x streams; utterances cut to odd and even lengths; and the
    alignments against the oracle pipeline that carries the history across the rewind."""
    from tests.test_oracle_e2e_goforward import default_configuration_alignment
    O = oracle_mod
    from oracle import fsg_oracle as F
    gpu = gpu_en if name == "en-us" else gpu_fr
    d = os.path.join(MODEL_ROOT, name)
    m = O.Model(d)
    olex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    cep = np.load(os.path.join(ROOT, "tests", "golden", "goforward_mfcc.npy" if name == "en-us"
                               else "goforward_fr_mfcc.npy")).astype(np.float32)
    n = len(cep)
    texts = (["go forward ten meters", "go forward", "go forward ten meters", "ten"] if name == "en-us"
             else ["avance de dix mètres", "avance", "dix mètres avance de", "de de de"])
    lens = [n, n - 1, 131, 2]                       # even, odd, odd, two frames
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = gpu.feat_batch(np.concatenate([cep[:k] for k in lens]), utt_off=off)
    lex = _lex(gpu, name)
    d_feats = torch.from_numpy(feats).cuda()
    cfg = lex.first_pass_config(two_pass_history=1)
    aset = ssw.align_text_batch_active(gpu, lex, d_feats, off, [t.split() for t in texts], cfg=cfg)
    try:
        rows = gpu.first_pass_active_carry(len(texts))
        for u, t in enumerate(texts):
            f_u = feats[off[u]:off[u + 1]]
            m.ptm_reset()
            m.ptm_set_frame_idx(0)

            def eval_frame(f, feat, lst):
                row = m.ptm_frame_eval(feat, f, compallsen=False, senone_active=lst)
                m.ptm_set_frame_idx(f + 1)
                return row

            state = {}

            def rewind():                            # the scorer's state as the first pass left it
                T = len(f_u)
                last_odd = ((T >> 1) << 1) - 1
                state["cw"] = m.ptm_get_topn(last_odd)[0] if T >= 2 else None
                m.ptm_set_frame_idx(0)

            seg, ph_start, ph_dur, ph_score = default_configuration_alignment(
                O, m, f_u, eval_frame, rewind, text=t, model=name)
            if seg is None:
                assert aset.status(u) == 1, t
                continue
            assert state["cw"] is not None
            assert np.array_equal(rows[u].astype(np.int32), state["cw"]), (t, lens[u])
            if aset.status(u) != 0:
                continue
            a = aset.utterance(u)
            assert [int(e[0]) for e in a["phone_al"]] == [int(x) for x in ph_start], t
            assert [int(e[2]) for e in a["phone_al"]] == [int(x) for x in ph_score], t
    finally:
        aset.free()


def test_texts_in_one_batch_match_the_frame_synchronous_oracle(gpu_en, orc_en, oracle_mod):
    """The judge's seven en-us texts (one without a path) over goforward as ONE batch, with the
    matrix-core scan and with the vector-unit one; plus the same texts one by one."""
    O = oracle_mod
    F, olex = _olex(O, orc_en, "en-us")
    texts = [t.split() for t in REF_EN_TEXTS]
    feats, off, n = _cep_batch(gpu_en, "goforward_mfcc.npy", len(texts))
    lex = _lex(gpu_en, "en-us")
    d_feats = torch.from_numpy(feats).cuda()
    want = [oracle_default_first_pass(O, F, orc_en, olex, t, feats[:n])[0] for t in texts]
    assert sum(w is None for w in want) == 1
    before = gpu_en.first_pass_active_stats()
    segs, rounds = lex.first_pass_active(d_feats, off, texts)
    after = gpu_en.first_pass_active_stats()
    assert after[0] - before[0] == len(texts) and after[1] - before[1] == int(rounds.sum())
    print("rounds per utterance:", rounds.tolist())
    for t, g, w in zip(texts, segs, want):
        if w is None:
            assert g is None, t
        else:
            assert [(a, s, s + d - 1, sc) for (a, s, d, sc) in g] == w, t
    for u in (0, 3, len(texts) - 1):
        one, _ = lex.first_pass_active(d_feats[u * n:(u + 1) * n], off[:2], [texts[u]])
        assert one[0] == segs[u]
    os.environ["SSW_SCAN"] = "fma"
    try:
        segs_fma, _ = lex.first_pass_active(d_feats, off, texts)
    finally:
        del os.environ["SSW_SCAN"]
    assert segs_fma == segs
    # the rounds over the unproven utterances one by one (what large batches do once few are
    # left: their unlisted entries refreshed from their compallsen = yes scores), and never
    for sub in ("1", "0"):
        os.environ["SSW_FPA_SUB"] = sub
        try:
            segs_sub, rounds_sub = lex.first_pass_active(d_feats, off, texts)
        finally:
            del os.environ["SSW_FPA_SUB"]
        assert segs_sub == segs, sub
        print("SSW_FPA_SUB=%s rounds per utterance:" % sub, rounds_sub.tolist())


def test_fr_fr_texts_in_one_batch(gpu_fr, orc_fr, oracle_mod):
    O = oracle_mod
    F, olex = _olex(O, orc_fr, "fr-fr")
    texts = [t.split() for t in REF_FR_TEXTS]
    feats, off, n = _cep_batch(gpu_fr, "goforward_fr_mfcc.npy", len(texts))
    lex = _lex(gpu_fr, "fr-fr")
    d_feats = torch.from_numpy(feats).cuda()
    segs, rounds = lex.first_pass_active(d_feats, off, texts)
    print("rounds per utterance:", rounds.tolist())
    for t, g in zip(texts, segs):
        w = oracle_default_first_pass(O, F, orc_fr, olex, t, feats[:n])[0]
        if w is None:
            assert g is None, t
        else:
            assert [(a, s, s + d - 1, sc) for (a, s, d, sc) in g] == w, t


def test_ragged_batch_of_cut_recordings(gpu_en, orc_en, oracle_mod):
    """Utterances of different lengths (the recording cut short at several points, an empty one
    among them), texts that fit and texts that do not: each against the oracle on its own."""
    O = oracle_mod
    F, olex = _olex(O, orc_en, "en-us")
    cep = np.load(os.path.join(ROOT, "tests", "golden", "goforward_mfcc.npy")).astype(np.float32)
    lens = [278, 130, 0, 64, 213, 1]
    texts = ["go forward ten meters", "go forward", "go", "go forward ten meters", "forward ten",
             "ten"]
    texts = [t.split() for t in texts]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = gpu_en.feat_batch(np.concatenate([cep[:k] for k in lens]), utt_off=off)
    lex = _lex(gpu_en, "en-us")
    d_feats = torch.from_numpy(feats).cuda()
    d_rows = torch.zeros((int(off[-1]), gpu_en.n_sen), dtype=torch.int16, device="cuda")
    segs, rounds, seed = lex.first_pass_active(d_feats, off, texts, d_senscr=d_rows, want_seed=True)
    torch.cuda.synchronize()
    rows = d_rows.cpu().numpy()
    print("rounds per utterance:", rounds.tolist())
    n_ok = 0
    for u, t in enumerate(texts):
        a, b = off[u], off[u + 1]
        if a == b:
            assert segs[u] is None
            continue
        w, wrows, wvec = oracle_default_first_pass(O, F, orc_en, olex, t, feats[a:b])
        if w is None:
            assert segs[u] is None, u
        else:
            n_ok += 1
            assert [(x, s, s + d - 1, sc) for (x, s, d, sc) in segs[u]] == w, u
        # rows and the seed are the reference's as far as its search got (it stops scoring when
        # it has no HMM left; here every utterance keeps some to its last frame or the oracle
        # returned early with None)
        if w is not None:
            assert np.array_equal(rows[a:b], wrows), u
            assert np.array_equal(seed[u], wvec), u
    assert n_ok >= 3


@pytest.mark.timeout(900)
def test_long_texts_take_the_long_text_kernels(gpu_en, orc_en, oracle_mod, monkeypatch):
    """Texts beyond 1,024 phone-tree HMMs: the sliding-window search kernel (and, forced, the
    HBM-resident one) export their active sets as well -- only the words of a frame's row that
    hold an active HMM, the rows cleared beforehand.  Two texts of 130 words with synthetic audio
    that follows them and a short one in the same batch, against the frame-synchronous oracle."""
    from tools.bench_first_pass import build_workload
    O = oracle_mod
    F, olex = _olex(O, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    texts, feats, nodes = build_workload(ssw, gpu_en, lex, 2, 2600, 130, noise=0.4, seed=77)
    assert nodes > 1024
    short_t, short_f, _ = build_workload(ssw, gpu_en, lex, 1, 300, 4, noise=0.4, seed=78)
    texts = texts + short_t
    feats = np.concatenate([feats, short_f])
    off = np.array([0, 2600, 5200, 5500], np.int32)
    d_feats = torch.from_numpy(feats).cuda()
    want = [oracle_default_first_pass(O, F, orc_en, olex, t, feats[off[u]:off[u + 1]])
            for u, t in enumerate(texts)]
    assert all(w[0] is not None for w in want)
    for knobs in ({}, {"SSW_FP_WIN": "0"}, {"SSW_FP_WIN": "0", "SSW_FP_BAND": "0"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        segs, rounds, seed = lex.first_pass_active(d_feats, off, texts, max_seg=2048, want_seed=True)
        for k in knobs:
            monkeypatch.delenv(k)
        print(knobs, "rounds per utterance:", rounds.tolist())
        for u, (g, (w, _, wvec)) in enumerate(zip(segs, want)):
            assert [(a, s, s + d - 1, sc) for (a, s, d, sc) in g] == w, (knobs, u)
            assert np.array_equal(seed[u], wvec), (knobs, u)


def test_ms_scorer(oracle_mod, orc_fr, tmp_path):
    """The same through the ms scorer (src/ms_mgau.c:322-365: only the listed senones are
    evaluated and written, best of them subtracted with the clamp; no codebook bookkeeping, no
    history): fr-fr with synthesised mixture_weights, the French recording with three texts."""
    O = oracle_mod
    from oracle import fsg_oracle as F
    from tests.test_cabi_host import synth_mixw_from_sendump
    src = os.path.join(MODEL_ROOT, "fr-fr")
    mixw = str(tmp_path / "mixture_weights")
    synth_mixw_from_sendump(orc_fr, mixw)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"),
              tmat=os.path.join(src, "transition_matrices"), mixw=mixw)
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = O.Model(vars=os.path.join(src, "variances"), **kw)
    olex = F.Lexicon(o, os.path.join(src, "dict.txt"), os.path.join(src, "noisedict.txt"))
    lex = ssw.Lexicon(g, os.path.join(src, "dict.txt"), os.path.join(src, "noisedict.txt"))
    texts = [t.split() for t in list(REF_FR_TEXTS)[:3]]
    feats, off, n = _cep_batch(g, "goforward_fr_mfcc.npy", len(texts))
    d_feats = torch.from_numpy(feats).cuda()
    d_rows = torch.zeros((len(feats), g.n_sen), dtype=torch.int16, device="cuda")
    segs, rounds = lex.first_pass_active(d_feats, off, texts, scorer=ssw.SCORER_MS, d_senscr=d_rows)
    torch.cuda.synchronize()
    rows = d_rows.cpu().numpy()
    print("rounds per utterance:", rounds.tolist())
    n_ok = 0
    for u, t in enumerate(texts):
        listed = []

        real = O.flags2list                          # spy: which entries each frame lists

        def spy(vec, n_sen):
            lst = real(vec, n_sen)
            listed.append(np.cumsum(np.asarray(lst, np.int64)))
            return lst
        O.flags2list = spy
        try:
            w, wrows, _ = oracle_default_first_pass(O, F, o, olex, t, feats[:n], ms=True)
        finally:
            O.flags2list = real
        if w is None:
            assert segs[u] is None, t
            continue
        n_ok += 1
        assert [(x, s, s + d - 1, sc) for (x, s, d, sc) in segs[u]] == w, t
        for f, ids in enumerate(listed):
            assert np.array_equal(rows[off[u] + f, ids], wrows[f, ids]), (t, f)
    assert n_ok >= 2
