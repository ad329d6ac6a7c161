"""End-to-end pin of the oracle against the reference's own output.

SURVEY.md Appendix C records what the real SoundSwallower library printed for config 1:
tests/data/goforward.wav, text "go forward ten meters", en-us, compallsen=yes -- the phone-level
alignment `start+duration(score)` of all 18 phones and the word scores.  This test recomputes
them with nothing but the oracle: front end -> batch CMN + 1s_c_d_dd -> PTM scoring (first pass
over all frames, then the second pass after acmod_rewind WITHOUT a history reset, as
decoder_alignment does, src/decoder.c:786-793) -> alignment_populate (restated below from
src/ps_alignment.c:132-247 on top of the oracle's bin_mdef_phone_id_nearest) ->
state_align_search.  Every score is an exact int; any deviation in the density arithmetic, the
top-N state machine, the log-add chain, the Viterbi step or the backtrace shows up here.

tests/golden/goforward.raw is the reference's tests/data/goforward.raw (the same samples as
goforward.wav without its 44-byte header).
"""
import os

import numpy as np
import pytest

from tests.conftest import MODEL_ROOT, ROOT

# SURVEY.md Appendix C, compallsen=yes
REF_PHONES = ("SIL 0+46(-90) G 46+8(-183) OW 54+10(-202) F 64+14(-286) AO 78+6(-204) "
              "R 84+10(-161) W 94+7(-124) ER 101+11(-206) D 112+5(-81) T 117+15(-588) "
              "EH 132+9(-150) N 141+12(-439) M 153+6(-72) IY 159+12(-580) T 171+3(-250) "
              "ER 174+16(-281) Z 190+21(-626) SIL 211+67(-900)")
REF_WORDS = [("<sil>", 0, 46, -90), ("go", 46, 18, -385), ("forward", 64, 53, -1062),
             ("ten", 117, 36, -1177), ("meters", 153, 58, -1809), ("<sil>", 211, 67, -900)]


def _parse_ref():
    out = []
    toks = REF_PHONES.split()
    for name, rest in zip(toks[0::2], toks[1::2]):
        se, sc = rest.split("(")
        s, d = se.split("+")
        out.append((name, int(s), int(d), int(sc.rstrip(")"))))
    return out


def _pronunciations(words, model="en-us"):
    d = os.path.join(MODEL_ROOT, model)
    want = set(words)
    pron = {}
    for fn in ("dict.txt", "noisedict.txt"):
        with open(os.path.join(d, fn)) as fh:
            for line in fh:
                parts = line.split()
                if parts and parts[0] in want and parts[0] not in pron:
                    pron[parts[0]] = parts[1:]
    return pron


def populate(O, m, words, model="en-us"):
    """alignment_populate (src/ps_alignment.c:132-247): words [(name, start, dur)] -> per phone
    (ciname, ssid, tmat, word index)."""
    pron = _pronunciations([w[0] for w in words], model)
    ci = lambda n: O.ciphone_id(m, n)
    sil = ci("SIL")
    ssid_of, tmat_of = m.phone_ssid, m.phone_tmat
    phones = []
    lc = sil
    for i, (w, _, _) in enumerate(words):
        p = [ci(x) for x in pron[w]]
        assert min(p) >= 0
        rc = ci(pron[words[i + 1][0]][0]) if i < len(words) - 1 else sil
        n = len(p)
        if n == 1:   # dict2pid_lrdiph_rc
            pid = O.phone_id_nearest(m, p[0], lc, rc, 3)
            phones.append((pron[w][0], int(ssid_of[pid]), int(tmat_of[p[0]]), i))
        else:        # dict2pid_ldiph_lc
            pid = O.phone_id_nearest(m, p[0], lc, p[1], 1)
            phones.append((pron[w][0], int(ssid_of[pid]), int(tmat_of[p[0]]), i))
            for j in range(1, n - 1):   # dict2pid_internal
                pid = O.phone_id_nearest(m, p[j], p[j - 1], p[j + 1], 0)
                phones.append((pron[w][j], int(ssid_of[pid]), int(tmat_of[p[j]]), i))
            pid = O.phone_id_nearest(m, p[-1], p[-2], rc, 2)   # rssid[...][cimap[rc]]
            phones.append((pron[w][-1], int(ssid_of[pid]), int(tmat_of[p[-1]]), i))
        lc = p[-1]
    return phones


def goforward_features(O):
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")
    assert len(pcm) == 44580
    # model/en-us/feat_params.json
    cep = O.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                    transform="dct")
    assert cep.shape == (278, 13)            # decoder_n_frames = 279 = output_frame + 1
    return O.feat_1s_c_d_dd(cep)


def two_pass_scores(m, feats):
    """First pass scores every frame once; decoder_alignment then rewinds (frame_idx = 0) and
    scores every frame again, carrying the top-N history over (no reset)."""
    m.ptm_score_utt(feats)
    m.ptm_set_frame_idx(0)
    out = np.zeros((len(feats), m.n_sen), np.int16)
    for t in range(len(feats)):
        out[t] = m.ptm_frame_eval(feats[t], t)
        m.ptm_set_frame_idx(t + 1)
    return out


def goforward_alignment_inputs(O, m):
    words = [(w, s, d) for (w, s, d, _) in REF_WORDS]
    phones = populate(O, m, words)
    ssid = np.array([p[1] for p in phones])
    senid = m.sseq[ssid]
    tmat = np.array([p[2] for p in phones], np.int16)
    wstart = np.array([words[p[3]][1] for p in phones], np.int32)
    wdur = np.array([words[p[3]][2] for p in phones], np.int32)
    sf = np.where(wstart > 0, wstart, 0).astype(np.int32)      # state_align_search.c:464-467
    ef = np.where(wdur > 0, wstart + wdur, 2**31 - 1).astype(np.int32)
    state_init = np.stack([np.repeat(wstart, 3), np.repeat(wdur, 3),
                           np.zeros(3 * len(phones), np.int32)], 1).astype(np.int32)
    return phones, senid, tmat, sf, ef, state_init


def test_front_end_matches_reference_golden_table(oracle_mod):
    """tests/_test_fe.res lines 11-15: MFCCs of the first 1024 samples with the default front
    end, printed with two decimals by tests/test_fe.c."""
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")[:1024]
    golden = """5.31 -0.58 -0.21 -0.00 -0.02 -0.08 -0.11 -0.05 0.09 0.01 -0.09 -0.20 -0.08
5.19 -0.54 -0.16 -0.15 0.02 0.09 -0.06 -0.07 -0.08 -0.13 -0.24 -0.17 0.14
5.19 -0.64 -0.20 -0.35 -0.03 -0.15 0.04 0.01 -0.14 -0.02 0.01 -0.05 0.02
5.36 -0.47 -0.17 -0.21 -0.12 -0.12 -0.08 -0.04 -0.05 0.00 -0.02 0.01 -0.01
5.21 -0.58 -0.15 -0.05 -0.16 -0.13 -0.17 -0.20 -0.16 -0.04 -0.00 0.03 -0.08"""
    want = [ln.split() for ln in golden.splitlines()]
    cep = oracle_mod.fe_mfcc(pcm)
    assert cep.shape == (5, 13)
    got = [["%.2f" % v for v in row] for row in cep]
    assert got == want


def test_goforward_alignment_matches_reference_output(oracle_mod, orc_en):
    O, m = oracle_mod, orc_en
    feats = goforward_features(O)
    scr = two_pass_scores(m, feats)
    phones, senid, tmat, sf, ef, state_init = goforward_alignment_inputs(O, m)
    ref = _parse_ref()
    assert [p[0] for p in phones] == [r[0] for r in ref]      # js/tests.js:127-130 phone string
    rv, st, ph = m.state_align(scr, senid, tmat, sf=sf, ef=ef, state_init=state_init)
    assert rv == 0
    got = [(phones[i][0], int(ph[i, 0]), int(ph[i, 1]), int(ph[i, 2])) for i in range(len(ph))]
    assert got == ref
    parent = np.array([p[3] for p in phones])
    words = [(int(ph[parent == w, 0][0]), int(ph[parent == w, 1].sum()), int(ph[parent == w, 2].sum()))
             for w in range(len(REF_WORDS))]
    assert words == [(s, d, sc) for (_, s, d, sc) in REF_WORDS]


def test_history_sensitivity_of_goforward(oracle_mod, orc_en):
    """SURVEY 8(c) fixture 4: on goforward the survey found exactly one frame whose PTM scores
    depend on the carried top-N history (sequential scoring differs from scoring the same
    frame from the reset history).  The restatement shows the same count, and the GPU tests
    check both semantics against it."""
    feats = goforward_features(oracle_mod)
    orc_en.ptm_reset()
    seq = orc_en.ptm_score_utt(feats)                  # history carried frame to frame
    differ = []
    for t in range(len(feats)):
        orc_en.ptm_reset()
        orc_en.ptm_set_frame_idx(0)
        fresh = orc_en.ptm_frame_eval(feats[t], 0)
        if not np.array_equal(fresh, seq[t]):
            differ.append(t)
    orc_en.ptm_reset()
    assert len(differ) == 1, differ


# SURVEY.md Appendix C, compallsen=no (the reference's DEFAULT configuration): same boundaries,
# these phone scores
REF_SCORES_DEFAULT = [-67, -49, -56, -100, -47, -54, -43, -59, -35, -207, -53, -127, -42, -65,
                      -136, -67, -141, -379]


def default_configuration_alignment(O, m, feats, eval_frame, rewind,
                                    text="go forward ten meters", model="en-us"):
    """The reference's default is compallsen=no: acmod scores only the senones of the active
    HMMs, through the uint8 delta list of acmod_flags2list (src/acmod.c:947-999), and the scorer
    normalises over that set.  The first pass clears and rebuilds the set every frame
    (fsg_search_sen_active); the second pass never clears it (state_align_search_step only
    activates, src/state_align_search.c:185-188), so it starts from the first pass's last set
    and grows; the top-N history carries over the rewind.  Both searches are restated here
    around a per-frame scorer `eval_frame(frame, feature row, delta list) -> int16 [n_sen]`
    (which also advances the scorer's frame index, as acmod_advance does); `rewind()` is
    acmod_rewind.  Returns (first-pass segmentation, phone start, duration, score)."""
    import ctypes as C
    from oracle import fsg_oracle as F
    d = os.path.join(MODEL_ROOT, model)
    lex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    T = len(feats)
    n_words32 = (m.n_sen + 31) // 32
    vec = np.zeros(n_words32, np.uint32)

    def set_bits(sen):
        for s_ in np.unique(np.asarray(sen).ravel()):
            vec[int(s_) >> 5] |= np.uint32(1 << (int(s_) & 31))

    def score(f):
        return eval_frame(f, feats[f], O.flags2list(vec, m.n_sen))

    # ---- first pass: the active set is rebuilt from the active HMMs every frame
    def first_pass_scores(f, sen):
        vec[:] = 0
        set_bits(sen)
        return score(f)

    seg = F.first_pass(m, lex, text.split(), first_pass_scores, n_frames=T)
    if seg is None:
        return None, None, None, None
    # ---- second pass (decoder_alignment): rewind, constrained windows, growing active set
    words = [(w, s, e - s + 1) for (w, s, e, _) in seg]
    phones = populate(O, m, words, model)
    n = len(phones)
    senid = np.ascontiguousarray(m.sseq[[p[1] for p in phones]], np.uint16)
    tmat = np.array([p[2] for p in phones], np.int16)
    wstart = np.array([words[p[3]][1] for p in phones])
    wdur = np.array([words[p[3]][2] for p in phones])
    sf = np.where(wstart > 0, wstart, 0)
    ef = np.where(wdur > 0, wstart + wdur, 2**31 - 1)
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
    rewind()                               # acmod_rewind
    sc[0, 0], hi[0, 0], frame_of[0] = 0, 0, 0     # state_align_search_start: hmm_enter(hmms, 0, 0, 0)
    tokens = np.full((T, n * 3, 2), -1, np.int64)
    for f in range(T):
        act = np.nonzero(frame_of == f)[0]
        set_bits(senid[act])               # activate only: the set keeps every earlier bit
        row = score(f)
        idx = np.nonzero(frame_of >= f)[0].astype(np.int32)
        L.orc_hmm_vit_eval_many(m._m, row.ctypes.data, len(idx), idx.ctypes.data, senid.ctypes.data,
                                tmat.ctypes.data, sc.ctypes.data, hi.ctypes.data, osc.ctypes.data,
                                ohi.ctypes.data, best.ctypes.data)
        nf = f + 1
        for i in idx:                      # prune_hmms
            if nf <= ef[i]:
                frame_of[i] = nf
        for i in range(n - 1):             # phone_transition
            if frame_of[i] != nf or nf < sf[i + 1]:
                continue
            if frame_of[i + 1] < f or osc[i] > sc[i + 1, 0]:
                sc[i + 1, 0], hi[i + 1, 0], frame_of[i + 1] = osc[i], ohi[i], nf
        for i in np.nonzero(frame_of >= f)[0]:      # record_transitions
            for j in range(3):
                tokens[f, i * 3 + j] = (hi[i, j], sc[i, j])
                hi[i, j] = i * 3 + j
    # state_align_search_finish
    last_id, last_sc = int(ohi[n - 1]), int(osc[n - 1])
    assert last_id != -1
    cur_id = last_id
    last_frame = T
    st = np.zeros((n * 3, 3), np.int64)
    for cf in range(T - 2, -1, -1):
        cid, csc = tokens[cf, cur_id]
        assert cid != -1
        if cid != last_id:
            st[last_id] = (cf + 1, last_frame - (cf + 1), last_sc - csc)
            last_id, last_sc, last_frame = int(cid), int(csc), cf + 1
        cur_id = int(cid)
    st[0, 0], st[0, 1] = 0, last_frame
    ph_start = st[0::3, 0]
    ph_dur = st.reshape(n, 3, 3)[:, :, 1].sum(1)
    ph_score = st.reshape(n, 3, 3)[:, :, 2].sum(1)
    return seg, ph_start, ph_dur, ph_score


def test_default_configuration_scores(oracle_mod):
    """Appendix C records the phone scores the real library printed in its default
    configuration (compallsen=no); the oracle's per-frame scorer with active lists, driven by
    the restated searches, reproduces them."""
    O = oracle_mod
    m = O.Model(os.path.join(MODEL_ROOT, "en-us"))
    feats = goforward_features(O)
    m.ptm_reset()
    m.ptm_set_frame_idx(0)

    def eval_frame(f, feat, lst):
        row = m.ptm_frame_eval(feat, f, compallsen=False, senone_active=lst)
        m.ptm_set_frame_idx(f + 1)
        return row

    seg, ph_start, ph_dur, ph_score = default_configuration_alignment(
        O, m, feats, eval_frame, lambda: m.ptm_set_frame_idx(0))
    assert [(w, s, e - s + 1) for (w, s, e, _) in seg] == [(w, s, dd) for (w, s, dd, _) in REF_WORDS]
    ref = _parse_ref()
    assert [(int(a), int(b)) for a, b in zip(ph_start, ph_dur)] == [(r[1], r[2]) for r in ref]
    assert [int(x) for x in ph_score] == REF_SCORES_DEFAULT


@pytest.mark.parametrize("text", ["go ten meters forward", "hello world", "ten",
                                  "fr:dix mètres avance de"])
def test_default_configuration_history_across_the_rewind(oracle_mod, text):
    """decoder_alignment's second pass starts from the top-N history the first pass left
    (acmod_rewind keeps it, src/decoder.c:786-793); ssw_align_text_batch_active starts each pass
    from the reset history (include/ssw_amd.h).  On the reference's recording the two agree for
    every text tried -- a carried order only matters where truncated densities tie, and the
    default configuration scores a few hundred senones a frame, not 5126 -- which is what this
    pins: words, phone boundaries and phone scores, carried against reset."""
    O = oracle_mod
    model = "fr-fr" if text.startswith("fr:") else "en-us"
    text = text[3:] if text.startswith("fr:") else text
    m = O.Model(os.path.join(MODEL_ROOT, model))
    if model == "fr-fr":
        from tests.test_first_pass_oracle import features
        feats = features(O, "goforward_fr.raw")
    else:
        feats = goforward_features(O)

    def run(carry):
        m.ptm_reset()
        m.ptm_set_frame_idx(0)

        def eval_frame(f, feat, lst):
            row = m.ptm_frame_eval(feat, f, compallsen=False, senone_active=lst)
            m.ptm_set_frame_idx(f + 1)
            return row

        def rewind():
            if not carry:
                m.ptm_reset()
            m.ptm_set_frame_idx(0)

        return default_configuration_alignment(O, m, feats, eval_frame, rewind, text=text,
                                               model=model)

    a, b = run(True), run(False)
    assert a[0] is not None and a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
