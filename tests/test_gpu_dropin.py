"""GPU: the vtable-shaped objects behave like the reference objects they stand in for."""
import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import lcg_uniform, synth_alignment_task, synth_features

pytestmark = pytest.mark.gpu


def test_mgau_vtable_frame_by_frame_matches_reference_semantics(gpu_en, orc_en, means_en):
    """ptm_mgau_frame_eval driven as acmod drives it: frame_idx written from outside, history
    carried from frame to frame AND across utterances (no reset), past frames re-scored from the
    stored top-N (src/ptm_mgau.c:425-448, src/acmod.c:367,760)."""
    g = ssw.PtmMgau(gpu_en)
    assert g.name == "ptm"
    assert g.transform() == -1
    orc_en.ptm_reset()
    for utt in range(2):                      # second utterance starts from the carried history
        feats = synth_features(means_en, 9, 500 + utt)
        g.frame_idx = 0
        orc_en.ptm_set_frame_idx(0)
        for t in range(len(feats)):
            got = g.frame_eval(feats[t], t)
            ref = orc_en.ptm_frame_eval(feats[t], t)
            assert np.array_equal(got, ref), (utt, t)
            if t == 4:                        # re-score a past frame: features are ignored
                g.frame_idx = t + 1
                orc_en.ptm_set_frame_idx(t + 1)
                again = g.frame_eval(np.zeros(39, np.float32), t)
                assert np.array_equal(again, ref)
            g.frame_idx = t + 1               # acmod_advance
            orc_en.ptm_set_frame_idx(t + 1)
    g.free()


def test_mgau_prescore_then_row_copies(gpu_en, orc_en, means_en):
    g = ssw.PtmMgau(gpu_en)
    feats = synth_features(means_en, 50, 77)
    g.prescore(feats)
    ref = orc_en.ptm_score_utt(feats)
    for t in (0, 17, 49):
        assert np.array_equal(g.frame_eval(feats[t], t), ref[t])
    g.free()


def test_state_align_search_object(gpu_en, orc_en, means_en):
    """init/start/step/finish as decoder_alignment drives them (src/decoder.c:777-795)."""
    n_ph, n_fr = 9, 80
    senid, tmat, ssid = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                             orc_en.n_ciphone, n_ph, 31)
    feats = synth_features(means_en, n_fr, 31)
    g = ssw.PtmMgau(gpu_en)
    s = ssw.StateAlignSearch(gpu_en, g, ssid, tmat)
    s.start()
    with pytest.raises(ssw.SswError, match="out of order"):
        s.step(feats[1], 1)
    for t in range(n_fr):
        s.step(feats[t], t)
    s.finish()
    scr = orc_en.ptm_score_utt(feats)
    rv, rst, rph = orc_en.state_align(scr, senid, tmat)
    assert rv == 0
    assert np.array_equal(s.states(), rst)
    assert np.array_equal(s.phones(), rph)
    s.free()
    g.free()


def test_state_align_failure_message(gpu_en, orc_en, means_en):
    senid, tmat, ssid = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                             orc_en.n_ciphone, 10, 3)
    feats = synth_features(means_en, 6, 3)
    s = ssw.StateAlignSearch(gpu_en, None, ssid, tmat)
    s.start()
    for t in range(6):
        s.step(feats[t], t)
    with pytest.raises(ssw.SswError, match="Failed to reach final state"):
        s.finish()


def test_align_batch_ragged_with_constraints(gpu_en, orc_en):
    """Several utterances of different shapes in one launch, word windows (sf/ef) on one."""
    shapes = [(3, 30), (70, 400), (1, 1), (20, 25), (130, 500)]
    frame_off, phone_off = [0], [0]
    senids, tmats, sfs, efs, scrs, refs = [], [], [], [], [], []
    for i, (n_ph, n_fr) in enumerate(shapes):
        senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                              orc_en.n_ciphone, n_ph, 1000 + i)
        scr = np.floor(lcg_uniform(2000 + i, n_fr * orc_en.n_sen).reshape(n_fr, -1) * 700).astype(
            np.int16)
        sf = np.zeros(n_ph, np.int32)
        ef = np.full(n_ph, 2**31 - 1, np.int32)
        if i == 1:
            sf[35:] = 200
            ef[:35] = 200
        refs.append(orc_en.state_align(scr, senid, tmat, sf=sf, ef=ef))
        senids.append(senid); tmats.append(tmat); sfs.append(sf); efs.append(ef); scrs.append(scr)
        frame_off.append(frame_off[-1] + n_fr)
        phone_off.append(phone_off[-1] + n_ph)
    d = gpu_en.to_device(np.concatenate(scrs))
    try:
        st, status = gpu_en.align_batch(d, frame_off, phone_off, np.concatenate(senids),
                                        np.concatenate(tmats), np.concatenate(sfs),
                                        np.concatenate(efs))
    finally:
        gpu_en.device_free(d)
    for i, (rv, rst, _) in enumerate(refs):
        assert (status[i] == 0) == (rv == 0), i
        if rv == 0:
            assert np.array_equal(st[phone_off[i] * 3:phone_off[i + 1] * 3], rst), i
    assert status[2] != 0 or shapes[2] == (1, 1)


def test_renormalisation_path(gpu_en, orc_en):
    """Scores large enough to trip `best_score - 0x300000 < WORST_SCORE`
    (src/state_align_search.c:193-197) on a long utterance."""
    n_ph, n_fr = 4, 17500
    senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                          orc_en.n_ciphone, n_ph, 77)
    scr = np.full((n_fr, orc_en.n_sen), 32000, np.int16)
    scr[:, ::7] = 31000
    rv, rst, _, trace = orc_en.state_align(scr, senid, tmat, want_trace=True)
    assert trace.min() - 0x300000 < -536870912, "test must actually reach the renormalisation"
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, [0, n_fr], [0, n_ph], senid, tmat)
    finally:
        gpu_en.device_free(d)
    assert (status[0] == 0) == (rv == 0)
    if rv == 0:
        assert np.array_equal(st, rst)


def test_goforward_end_to_end_matches_reference_output_on_gpu(gpu_en, orc_en, oracle_mod):
    """Config 1 through the GPU: features of goforward (oracle front end) -> the mgau vtable
    object scoring both passes frame by frame with the history carried across the rewind, as
    decoder_alignment does -> ssw_align_batch with the word windows of the first pass.  Must
    reproduce the reference library's recorded phone alignment exactly."""
    from tests.test_oracle_e2e_goforward import (REF_WORDS, _parse_ref, goforward_alignment_inputs,
                                                 goforward_features)
    feats = goforward_features(oracle_mod)
    g = ssw.PtmMgau(gpu_en)
    n = len(feats)
    for t in range(n):                       # first pass
        g.frame_eval(feats[t], t)
        g.frame_idx = t + 1
    g.frame_idx = 0                          # acmod_rewind
    scr = np.zeros((n, gpu_en.n_sen), np.int16)
    for t in range(n):                       # second pass
        scr[t] = g.frame_eval(feats[t], t)
        g.frame_idx = t + 1
    g.free()
    phones, senid, tmat, sf, ef, state_init = goforward_alignment_inputs(oracle_mod, orc_en)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, [0, n], [0, len(phones)], senid, tmat, sf, ef,
                                        state_init)
    finally:
        gpu_en.device_free(d)
    assert status[0] == 0
    ph = gpu_en.propagate(st, np.arange(len(st)) // 3, len(phones))
    got = [(phones[i][0], int(ph[i, 0]), int(ph[i, 1]), int(ph[i, 2])) for i in range(len(ph))]
    assert got == _parse_ref()
    words = gpu_en.propagate(ph, [p[3] for p in phones], len(REF_WORDS))
    assert [tuple(int(x) for x in w) for w in words] == [(s, dd, sc) for (_, s, dd, sc) in REF_WORDS]
    # the batched path (history reset at the utterance start) gives the same alignment here
    scr2 = gpu_en.score_batch(feats)
    d = gpu_en.to_device(scr2)
    try:
        st2, status2 = gpu_en.align_batch(d, [0, n], [0, len(phones)], senid, tmat, sf, ef,
                                          state_init)
    finally:
        gpu_en.device_free(d)
    assert status2[0] == 0 and np.array_equal(st2[:, :2], st[:, :2])


def _random_active_list(oracle_mod, n_sen, rng, density):
    vec = np.zeros((n_sen + 31) // 32, np.uint32)
    sens = np.flatnonzero(rng.random(n_sen) < density)
    for s in sens:
        vec[s // 32] |= np.uint32(1 << (s % 32))
    return oracle_mod.flags2list(vec, n_sen)


def test_mgau_vtable_compallsen_no(gpu_en, orc_en, oracle_mod, means_en):
    """frame_eval with an active-senone delta list (src/ptm_mgau.c:297-321, 353-364, 392-400):
    codebook activity follows the list, inactive codebooks only re-score their carried
    codewords, the normaliser and the best score run over the active set, and the best score is
    subtracted from every entry.  Sparse lists exercise the >255 bridge entries."""
    rng = np.random.default_rng(5)
    g = ssw.PtmMgau(gpu_en)
    orc_en.ptm_reset()
    feats = synth_features(means_en, 14, 321)
    for t in range(len(feats)):
        dens = (0.002, 0.02, 0.3, 1.0)[t % 4]
        lst = _random_active_list(oracle_mod, orc_en.n_sen, rng, dens)
        if t == 5:
            lst = lst[:0]                                   # nothing active at all
        got = g.frame_eval(feats[t], t, compallsen=False, senone_active=lst)
        ref = orc_en.ptm_frame_eval(feats[t], t, compallsen=False, senone_active=lst)
        assert np.array_equal(got, ref), t
        if t == 8:      # mix in an all-senone frame, and re-score a past frame with a new list
            g.frame_idx = t + 1
            orc_en.ptm_set_frame_idx(t + 1)
            lst2 = _random_active_list(oracle_mod, orc_en.n_sen, rng, 0.1)
            again = g.frame_eval(feats[t], t, compallsen=False, senone_active=lst2)
            ref2 = orc_en.ptm_frame_eval(feats[t], t, compallsen=False, senone_active=lst2)
            assert np.array_equal(again, ref2)
        g.frame_idx = t + 1
        orc_en.ptm_set_frame_idx(t + 1)
    g.free()


def test_goforward_from_words_with_product_glue(gpu_en, oracle_mod):
    """Words + word windows -> Lexicon.populate (host C) -> StateAlignSearch (GPU), features
    from the device feature kernel: the product's own pieces end to end, against the reference's
    recorded alignment."""
    import os
    from tests.conftest import MODEL_ROOT, ROOT
    from tests.test_oracle_e2e_goforward import REF_WORDS, _parse_ref
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")
    cep = oracle_mod.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                             transform="dct")         # the MFCC front end is outside the path
    feats = gpu_en.feat_batch(cep)
    d = os.path.join(MODEL_ROOT, "en-us")
    lex = ssw.Lexicon(gpu_en, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    rows = lex.populate([w for (w, _, _, _) in REF_WORDS], [s for (_, s, _, _) in REF_WORDS],
                        [dd for (_, _, dd, _) in REF_WORDS])
    g = ssw.PtmMgau(gpu_en)
    s = ssw.StateAlignSearch(gpu_en, g, rows["ssid"], rows["tmatid"], rows["start"],
                             rows["duration"])
    s.start()
    for t in range(len(feats)):
        s.step(feats[t], t)
    s.finish()
    ph = s.phones()
    ref = _parse_ref()
    # boundaries must equal the reference's; scores too, because the batch scorer's reset
    # history does not change any senone score on this utterance
    assert [(int(a), int(b)) for a, b, _ in ph] == [(r[1], r[2]) for r in ref]
    assert [int(c) for _, _, c in ph] == [r[3] for r in ref]
    # ... and printed the reference's way it is the reference's line (decoder_result_json,
    # align_level 1, as recorded in SURVEY Appendix C)
    from tests.test_lexicon_host import REF_JSON_PREFIX
    words = [w for (w, _, _, _) in REF_WORDS]
    word_al = gpu_en.propagate(ph, rows["parent"], len(words))
    line = lex.alignment_json("go forward ten meters", words, word_al, rows["cipid"],
                              rows["parent"], ph, n_frames=279)  # decoder_n_frames, Appendix C
    assert line.startswith(REF_JSON_PREFIX)
    s.free()
    g.free()
    lex.free()


def test_default_configuration_scores_through_the_gpu_scorer(gpu_en, orc_en, oracle_mod):
    """The reference's DEFAULT configuration (compallsen=no) end to end with the GPU scorer: every
    frame of both passes is scored by vt->frame_eval with the active senone list acmod would
    pass (rebuilt per frame in the first pass, only growing in the second), the top-N history
    carried over the rewind; the searches are the restated ones of the oracle test.  The phone
    scores must be the ones the real library printed (compallsen=no)."""
    from tests.test_oracle_e2e_goforward import (REF_SCORES_DEFAULT, REF_WORDS, _parse_ref,
                                                 default_configuration_alignment,
                                                 goforward_features)
    feats = goforward_features(oracle_mod)
    g = ssw.PtmMgau(gpu_en)
    g.reset_hist()
    g.frame_idx = 0

    def eval_frame(f, feat, lst):
        row = g.frame_eval(feat, f, compallsen=False, senone_active=lst)
        g.frame_idx = f + 1              # acmod_advance
        return row

    def rewind():
        g.frame_idx = 0                  # acmod_rewind

    seg, ph_start, ph_dur, ph_score = default_configuration_alignment(
        oracle_mod, orc_en, feats, eval_frame, rewind)
    assert [(w, s, e - s + 1) for (w, s, e, _) in seg] == [(w, s, d) for (w, s, d, _) in REF_WORDS]
    ref = _parse_ref()
    assert [(int(a), int(b)) for a, b in zip(ph_start, ph_dur)] == [(r[1], r[2]) for r in ref]
    assert [int(x) for x in ph_score] == REF_SCORES_DEFAULT
    g.free()


def test_second_pass_scores_from_the_batch_api(oracle_mod, orc_en, gpu_en):
    """decoder_alignment rewinds and scores every frame again WITHOUT resetting the top-N history
    (src/decoder.c:786-793): the batch API reproduces the second pass's scores on all 278 frames
    of the reference's recording when the second call starts from the first call's carry_out
    (history-dependent frames in quantity: tests/test_gpu_ptm.py,
    test_history_carried_between_calls_and_utterances)."""
    from tests.test_oracle_e2e_goforward import goforward_features, two_pass_scores
    feats = goforward_features(oracle_mod)
    first, carry = gpu_en.score_batch_carry(feats)
    second, _ = gpu_en.score_batch_carry(feats, carry_in=carry)
    ref_first = orc_en.ptm_score_utt(feats)
    ref_second = two_pass_scores(orc_en, feats)
    assert np.array_equal(first, ref_first)
    assert np.array_equal(second, ref_second)


def test_a_model_refuses_a_second_thread(gpu_en, means_en):
    """include/ssw_amd.h, Threading: an ssw_model_t is not re-entrant; round 3 enforces it.  While
    one thread scores large batches another thread's calls on the SAME model fail with "in use by
    another thread" (and never corrupt the first thread's results); on a model of its own the
    second thread works."""
    import threading
    feats = np.concatenate([synth_features(means_en, 256, 9000 + u) for u in range(128)])
    off = (np.arange(129) * 256).astype(np.int32)
    want = gpu_en.score_batch(feats, off)
    stop, refused, other_ok = threading.Event(), [], []

    def intruder():
        small = synth_features(means_en, 40, 1)
        while not stop.is_set():
            try:
                gpu_en.score_batch(small)
                other_ok.append(1)
            except ssw.SswError as e:
                refused.append(str(e))

    th = threading.Thread(target=intruder)
    th.start()
    try:
        for _ in range(6):
            got = gpu_en.score_batch(feats, off)      # a refusal of THIS thread would raise here
            assert np.array_equal(got, want)
    except ssw.SswError as e:                          # the intruder got in first: legitimate,
        assert "another thread" in str(e)              # the rule cuts both ways
    finally:
        stop.set()
        th.join()
    assert refused and all("another thread" in r for r in refused), (len(refused), len(other_ok))
    mine = ssw.Model(ssw.model_dir("en-us"))
    assert np.array_equal(mine.score_batch(feats[:256]), want[:256])
    mine.close()
