"""First pass of forced alignment on the GPU (ssw_first_pass_batch) against the oracle's
restatement of fsg_search (oracle/fsg_oracle.py) and against the word segmentations the
reference itself printed."""
import os

import numpy as np
import pytest
import torch

import soundswallower_amd as ssw
from soundswallower_amd.synth import lcg_uniform
from tests.conftest import MODEL_ROOT
from tests.test_first_pass_oracle import REF_EN, REF_FR, features

pytestmark = pytest.mark.gpu


def _lex(model, name):
    d = os.path.join(MODEL_ROOT, name)
    return ssw.Lexicon(model, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))


def _olex(oracle_mod, m, name):
    from oracle import fsg_oracle as F
    d = os.path.join(MODEL_ROOT, name)
    return F, F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))


def _first_pass(model, lex, senscr_list, texts):
    """scores (host int16 arrays, one per utterance) -> device -> ssw_first_pass_batch"""
    off = np.concatenate([[0], np.cumsum([len(s) for s in senscr_list])]).astype(np.int32)
    d = torch.from_numpy(np.ascontiguousarray(np.concatenate(senscr_list), np.int16)).cuda()
    return lex.first_pass(d, off, texts)


@pytest.mark.parametrize("name,raw,text,ref", [
    ("en-us", "goforward.raw", "go forward ten meters", REF_EN),
    ("fr-fr", "goforward_fr.raw", "avance de dix mètres", REF_FR)])
def test_reference_recordings(oracle_mod, gpu_en, gpu_fr, orc_en, orc_fr, name, raw, text, ref):
    gpu, orc = (gpu_en, orc_en) if name == "en-us" else (gpu_fr, orc_fr)
    feats = features(oracle_mod, raw)
    scr = gpu.score_batch(feats)
    lex = _lex(gpu, name)
    seg = _first_pass(gpu, lex, [scr], [text.split()])[0]
    assert [(w, s, d) for (w, s, d, _) in seg] == ref
    F, olex = _olex(oracle_mod, orc, name)
    want = F.first_pass(orc, olex, text.split(), scr)
    assert [(w, s, s + d - 1, sc) for (w, s, d, sc) in seg] == want


def synth_scores(F, orc, olex, words, seed, n_sen, noise_lo=120, noise_hi=400, sil_p=0.5,
                 cfg=None):
    """Senone scores that follow one path through the text's phone trees: per state a few frames
    where that state's senone scores near 0 and everything else is `noise`; optional silences
    between words; alternates picked at random."""
    lmath = O = None
    from oracle import oracle as O
    lmath = O.Logmath(1.0001, 0)
    cfg = cfg or F.Config
    arcs = F.build_fsg(olex, words, lmath, cfg)
    nodes, roots = F.build_lextree(orc, olex, arcs, 0, 0)
    rng = np.random.default_rng(seed)
    path = []
    def sil_node(s):
        r = roots[s]
        while r is not None:
            if r.leaf and r.link.word == "<sil>":
                return r
            r = r.sibling
    for s in range(len(arcs)):
        if cfg.fsgusefiller and (s == 0 or s == len(arcs) - 1 or rng.random() < sil_p):
            path.append(sil_node(s))
        if s == len(arcs) - 1:
            break
        cands = []
        r = roots[s]
        while r is not None:
            if not (r.leaf and r.link.filler):
                cands.append(r)
            r = r.sibling
        n = cands[rng.integers(len(cands))]
        while True:
            path.append(n)
            if n.leaf:
                break
            kids = []
            c = n.succ
            while c is not None:
                kids.append(c)
                c = c.sibling
            n = kids[rng.integers(len(kids))]
    rows = []
    for n in path:
        for st in range(3):
            for _ in range(int(rng.integers(1, 5))):
                row = rng.integers(noise_lo, noise_hi, n_sen).astype(np.int16)
                row[orc.sseq[n.ssid][st]] = rng.integers(0, 30)
                rows.append(row)
    return np.stack(rows)


def test_synthetic_paths_match_the_oracle(oracle_mod, gpu_en, orc_en):
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(7, 24 * 10)
    texts, scores = [], []
    for t in range(24):
        n = 1 + int(u[t * 10] * 8)
        words = [vocab[int(x * len(vocab))] for x in u[t * 10 + 1:t * 10 + 1 + n]]
        texts.append(words)
        # a few utterances get noise close to the path: near-ties, wrong turns, failures
        lo = 120 if t % 4 else 25
        scores.append(synth_scores(F, orc_en, olex, words, 100 + t, orc_en.n_sen, noise_lo=lo))
    got = _first_pass(gpu_en, lex, scores, texts)
    n_ok = 0
    for t in range(24):
        want = F.first_pass(orc_en, olex, texts[t], scores[t])
        if want is None:
            assert got[t] is None, texts[t]
            continue
        n_ok += 1
        assert got[t] is not None, texts[t]
        assert [(w, s, s + d - 1, sc) for (w, s, d, sc) in got[t]] == want, texts[t]
    assert n_ok >= 12


def test_wrong_text_and_short_audio(oracle_mod, gpu_en, orc_en):
    """The same recording against texts it does not contain, and truncated audio: whatever the
    reference's search does (a forced path, or no path into the final state) must come out."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    feats = features(oracle_mod, "goforward.raw")
    scr = gpu_en.score_batch(feats)
    texts = [t.split() for t in ("go forward", "ten meters go forward", "forward forward forward",
                                 "a", "go forward ten meters go forward ten meters")]
    cuts = [len(scr), len(scr), 120, 30, 200]
    got = _first_pass(gpu_en, lex, [scr[:c] for c in cuts], texts)
    for t, c, g in zip(texts, cuts, got):
        want = F.first_pass(orc_en, olex, t, scr[:c])
        if want is None:
            assert g is None, t
        else:
            assert [(w, s, s + d - 1, sc) for (w, s, d, sc) in g] == want, t


def test_text_and_audio_to_the_reference_json(oracle_mod, gpu_en):
    """decoder_set_align_text + process + decoder_alignment + decoder_result_json, the product's
    way: cepstra -> device features -> batch scores in HBM -> first pass -> populate with its
    word windows -> constrained state alignment -> JSON; no reference-derived windows anywhere.
    The line must be the one the reference printed; a second utterance in
    the batch (the same audio, a text that cannot be completed) must come back as None without
    disturbing the first."""
    from tests.conftest import ROOT
    from tests.test_lexicon_host import REF_JSON_PREFIX
    from tests.test_oracle_e2e_goforward import REF_WORDS, _parse_ref
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")
    cep = oracle_mod.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                             transform="dct")         # the MFCC front end is outside the path
    n = len(cep)
    feats = gpu_en.feat_batch(np.concatenate([cep, cep[:40]]), utt_off=[0, n, n + 40])
    d_feats = torch.from_numpy(feats).cuda()
    d_scr = torch.empty((n + 40, gpu_en.n_sen), dtype=torch.int16, device="cuda")
    off = np.array([0, n, n + 40], np.int32)
    gpu_en.score_batch_device(d_feats, n + 40, off, d_scr)
    torch.cuda.synchronize()
    lex = _lex(gpu_en, "en-us")
    texts = ["go forward ten meters".split(), "go forward ten meters".split()]
    res = ssw.forced_alignment(gpu_en, lex, d_scr, off, texts)
    assert res[1] is None                      # 40 frames cannot hold the sentence
    a = res[0]
    assert a["words"] == [w for (w, _, _, _) in REF_WORDS]
    assert [tuple(int(x) for x in r) for r in a["word_al"]] == [(s, d, sc) for (_, s, d, sc) in REF_WORDS]
    ref = _parse_ref()
    assert [tuple(int(x) for x in r) for r in a["phone_al"]] == [(r[1], r[2], r[3]) for r in ref]
    line = lex.alignment_json("go forward ten meters", a["words"], a["word_al"], a["cipid"],
                              a["parent"], a["phone_al"], n_frames=279)
    assert line.startswith(REF_JSON_PREFIX)
    # ... and straight from the C result object, hypothesis string and duration included
    aset = ssw.forced_align_batch(gpu_en, lex, d_scr, off, texts)
    assert aset.status(0) == 0 and aset.status(1) == 1
    assert aset.json(0) == line and aset.json(0, align_level=2).count('"t":"') > 3 * 18
    with pytest.raises(ssw.SswError, match="no alignment"):
        aset.json(1)
    aset.free()
    # ... and from the feature rows in one call (scores kept in the model's workspace)
    aset = ssw.align_text_batch(gpu_en, lex, d_feats, off, texts)
    assert aset.json(0) == line and aset.status(1) == 1
    aset.free()
    lex.free()


def test_long_texts_use_the_wider_kernels(oracle_mod, gpu_en, orc_en):
    """Texts of 60, 150 and 330 words: more than 512 / 1024 / 2048 phone-tree HMMs, i.e. the
    1024-thread register kernel and, beyond 1,024 HMMs, the sliding-window kernel (rounds 1-4:
    register instances with two and four HMMs per thread); same results as the oracle."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(31, 400)
    texts = [[vocab[int(x * len(vocab))] for x in u[:60]],
             [vocab[int(x * len(vocab))] for x in u[100:250]],
             [vocab[int(x * len(vocab))] for x in np.concatenate([u, u[::-1]])[:330]]]
    for t, least in zip(texts, (512, 1024, 2048)):
        assert len(lex.first_pass_graph(t)[0]) > least
    scores = [synth_scores(F, orc_en, olex, t, 5 + i, orc_en.n_sen, sil_p=0.2)
              for i, t in enumerate(texts)]
    got = _first_pass(gpu_en, lex, scores, texts)
    for t, sc, g in zip(texts, scores, got):
        want = F.first_pass(orc_en, olex, t, sc)
        assert want is not None and g is not None
        assert [(w, s, s + d - 1, x) for (w, s, d, x) in g] == want


def test_large_batch_equals_one_by_one(oracle_mod, gpu_en, orc_en):
    """64 utterances in one call (graphs built by several host threads and concatenated) give
    what 64 single-utterance calls give; an unknown word anywhere fails the call with the
    reference's message."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(77, 64 * 6)
    texts = [[vocab[int(x * len(vocab))] for x in u[t * 6:t * 6 + 1 + t % 5]] for t in range(64)]
    scores = [synth_scores(F, orc_en, olex, t, 900 + i, orc_en.n_sen) for i, t in enumerate(texts)]
    got = _first_pass(gpu_en, lex, scores, texts)
    for i in range(64):
        assert got[i] == _first_pass(gpu_en, lex, [scores[i]], [texts[i]])[0], i
    assert sum(g is not None for g in got) > 48
    # the same with the graphs prepared beforehand; a plan can be run again (other scores)
    off = np.concatenate([[0], np.cumsum([len(s) for s in scores])]).astype(np.int32)
    d = torch.from_numpy(np.ascontiguousarray(np.concatenate(scores), np.int16)).cuda()
    plan = ssw.FirstPassPlan(gpu_en, lex, texts)
    a1 = ssw.forced_align_planned(gpu_en, lex, plan, d, off)
    a2 = ssw.forced_align_batch(gpu_en, lex, d, off, texts)
    a3 = ssw.forced_align_planned(gpu_en, lex, plan, d, off)
    for i in range(64):
        u1, u2, u3 = a1.utterance(i), a2.utterance(i), a3.utterance(i)
        assert (u1 is None) == (u2 is None) == (u3 is None) == (got[i] is None)
        if u1 is not None:
            assert u1["words"] == u2["words"] == u3["words"] == [w for (w, _, _, _) in got[i]]
            assert np.array_equal(u1["state_al"], u2["state_al"]) and np.array_equal(u1["state_al"], u3["state_al"])
    for x in (a1, a2, a3):
        x.free()
    plan.free()
    bad = [list(t) for t in texts]
    bad[50] = bad[50] + ["qqqqq"]
    with pytest.raises(ssw.SswError, match="Unknown word qqqqq"):
        _first_pass(gpu_en, lex, scores, bad)
    # ... of the low-level call; decoder_set_align_text rejects that ONE utterance
    # (src/decoder.c:699-703) and so do the alignment calls: status 3 with the reference's
    # message, every other utterance as before
    d_feats_like = ssw.forced_align_batch(gpu_en, lex, d, off, bad)
    assert d_feats_like.status(50) == 3 and d_feats_like.message(50) == "Unknown word qqqqq"
    assert d_feats_like.utterance(50) is None and d_feats_like.message(49) == ""
    a2 = ssw.forced_align_batch(gpu_en, lex, d, off, texts)
    for i in range(64):
        if i == 50:
            continue
        u1, u2 = d_feats_like.utterance(i), a2.utterance(i)
        assert (u1 is None) == (u2 is None)
        if u1 is not None:
            assert u1["words"] == u2["words"] and np.array_equal(u1["state_al"], u2["state_al"])
    d_feats_like.free()
    a2.free()
    # a text with more phone-tree HMMs than a workgroup can hold goes through the kernel that
    # keeps them in HBM; a few frames of audio cannot reach the end of 800 words
    long_text = [vocab[int(x * len(vocab))] for x in lcg_uniform(5, 800)]
    assert len(lex.first_pass_graph(long_text)[0]) > 4096
    assert _first_pass(gpu_en, lex, [scores[0]], [long_text]) == [None]


def test_alternates_pronounced_alike(oracle_mod, gpu_fr, orc_fr):
    """fr-fr texts full of alternates with identical pronunciations (abus / abus(2), the
    one-phone ait / ait(2), ...): their HMMs tie for ever and the reference reports whichever
    its list order lets through at the exit frame; durations are varied so that exits fall on
    both parities."""
    F, olex = _olex(oracle_mod, orc_fr, "fr-fr")
    lex = _lex(gpu_fr, "fr-fr")
    texts, scores = [], []
    for t in range(40):
        words = [["abus", "ait", "mauritaniens"], ["ait", "abus"], ["mauritaniens", "abus", "abus"],
                 ["abus"]][t % 4]
        texts.append(words)
        scores.append(synth_scores(F, orc_fr, olex, words, 300 + t, orc_fr.n_sen, sil_p=0.5))
    got = _first_pass(gpu_fr, lex, scores, texts)
    seen = set()
    for t in range(40):
        want = F.first_pass(orc_fr, olex, texts[t], scores[t])
        assert want is not None and got[t] is not None
        assert [(w, s, s + d - 1, sc) for (w, s, d, sc) in got[t]] == want, texts[t]
        seen.update(w for (w, _, _, _) in got[t])
    assert {"abus", "abus(2)"} <= seen          # both members of a group do get reported


@pytest.mark.parametrize("kw", [
    dict(beam=1e-20, pbeam=1e-20, wbeam=1e-10),
    dict(use_filler=0),
    dict(use_altpron=0),
    dict(lw=9.5, wip=0.2, silprob=0.1, fillprob=1e-3),
    dict(beam=1e-80, pbeam=1e-60, wbeam=1e-40, pip=0.5),
])
def test_search_parameters_reach_the_search(oracle_mod, gpu_fr, orc_fr, kw):
    """beam / pbeam / wbeam, lw / wip / pip, silprob / fillprob, fsgusefiller, fsgusealtpron: the
    same values on both sides, same outcome (fr-fr: alternates matter there)."""
    F, olex = _olex(oracle_mod, orc_fr, "fr-fr")
    lex = _lex(gpu_fr, "fr-fr")
    names = {"use_filler": "fsgusefiller", "use_altpron": "fsgusealtpron"}
    ocfg = type("Cfg", (F.Config,), {names.get(k, k): (bool(v) if k in names else v)
                                     for k, v in kw.items()})
    cfg = lex.first_pass_config(**kw)
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(123, 12 * 8)
    texts = [[vocab[int(x * len(vocab))] for x in u[t * 8:t * 8 + 2 + t % 5]] for t in range(12)]
    texts[0] = ["avance", "de", "dix", "mètres"]
    scores = [synth_scores(F, orc_fr, olex, t, 40 + i, orc_fr.n_sen, noise_lo=60, cfg=ocfg)
              for i, t in enumerate(texts)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in scores])]).astype(np.int32)
    d = torch.from_numpy(np.ascontiguousarray(np.concatenate(scores), np.int16)).cuda()
    got = lex.first_pass(d, off, texts, cfg=cfg)
    n_ok = 0
    for t, sc, g in zip(texts, scores, got):
        want = F.first_pass(orc_fr, olex, t, sc, cfg=ocfg)
        if want is None:
            assert g is None, (kw, t)
        else:
            n_ok += 1
            assert g is not None and [(w, s, s + dd - 1, x) for (w, s, dd, x) in g] == want, (kw, t)
    assert n_ok >= 3


def test_edge_shapes(oracle_mod, gpu_en, orc_en):
    """An empty text (the grammar is one state with its filler loops), an utterance without
    frames in the middle of a batch, one-frame and three-frame audio, a one-word text."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    rng = np.random.default_rng(3)
    sil = synth_scores(F, orc_en, olex, [], 1, orc_en.n_sen)          # <sil> only
    one = synth_scores(F, orc_en, olex, ["go"], 2, orc_en.n_sen)
    texts = [[], ["go"], ["go"], ["go", "forward"], ["go"], []]
    scores = [sil, one, np.zeros((0, orc_en.n_sen), np.int16), one[:1], one[:3], one]
    import os
    # the register kernel, then (round 3) the long-text kernels forced on these shapes: the
    # sliding-window one and the HBM-resident one
    for env in ({}, {"SSW_FP_KERNEL": "big", "SSW_FP_WIN_TPB": "256"},
                {"SSW_FP_KERNEL": "big", "SSW_FP_WIN": "0"}):
        os.environ.update(env)
        try:
            got = _first_pass(gpu_en, lex, scores, texts)
        finally:
            for k in env:
                del os.environ[k]
        for t, sc, g in zip(texts, scores, got):
            want = F.first_pass(orc_en, olex, t, sc) if len(sc) else None
            if want is None:
                assert g is None, (t, len(sc), env)
            else:
                assert g is not None and [(w, s, s + d - 1, x) for (w, s, d, x) in g] == want, \
                    (t, len(sc), env)
        assert got[0] is not None and all(w == "<sil>" for (w, _, _, _) in got[0])
        assert got[1] is not None and got[2] is None


def test_full_size_properties(gpu_en):
    """BASELINE-sized utterances (1000 frames, texts of 25 words; 64 of them here) from features
    to alignments in one call, checked through what the domain guarantees whatever the scores:
    the words are the text's (fillers and alternates aside), word / phone / state entries tile
    the utterance without gaps, every phone lies inside its word and every state inside its
    phone, scores add up level by level (alignment_propagate)."""
    import sys
    from tests.conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_first_pass import build_workload
    lex = _lex(gpu_en, "en-us")
    utts, frames = 64, 1000
    texts, feats, _ = build_workload(ssw, gpu_en, lex, utts, frames, 25)
    off = (np.arange(utts + 1) * frames).astype(np.int32)
    aset = ssw.align_text_batch(gpu_en, lex, torch.from_numpy(feats).cuda(), off, texts)
    for u in range(utts):
        a = aset.utterance(u)
        assert a is not None, u
        spoken = [w.split("(")[0] for w in a["words"] if not w.startswith(("<", "["))]
        assert spoken == texts[u]
        for lvl in ("word_al", "phone_al", "state_al"):
            al = a[lvl]
            assert al[0, 0] == 0 and int(al[:, 1].sum()) == frames
            assert np.array_equal(al[1:, 0], al[:-1, 0] + al[:-1, 1])
        w, p, s = a["word_al"], a["phone_al"], a["state_al"]
        par = a["parent"]
        assert np.all(p[:, 0] >= w[par, 0]) and np.all(p[:, 0] + p[:, 1] <= w[par, 0] + w[par, 1])
        assert np.array_equal(np.bincount(par, weights=p[:, 2], minlength=len(w)).astype(np.int64),
                              w[:, 2].astype(np.int64))
        assert np.array_equal(s[:, 2].reshape(-1, 3).sum(1), p[:, 2])
        assert np.array_equal(s[0::3, 0], p[:, 0])
    aset.free()


def test_french_recording_from_cepstra(oracle_mod, gpu_fr):
    """The reference's second recording end to end (cepstra -> device features -> scores ->
    first pass -> populate -> constrained alignment): the words, with the alternates the
    reference picked, and their frames as it printed them (fr-fr PTM)."""
    from tests.conftest import ROOT
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward_fr.raw"), dtype="<i2")
    cep = oracle_mod.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                             transform="dct")
    feats = gpu_fr.feat_batch(cep)
    lex = _lex(gpu_fr, "fr-fr")
    off = np.array([0, len(feats)], np.int32)
    aset = ssw.align_text_batch(gpu_fr, lex, torch.from_numpy(feats).cuda(), off,
                                ["avance de dix mètres".split()])
    a = aset.utterance(0)
    assert [(w, int(r[0]), int(r[1])) for w, r in zip(a["words"], a["word_al"])] == REF_FR
    line = aset.json(0)
    assert line.startswith('{"b":0.000,"d":2.400,"p":1.000,"t":"avance de dix mètres","w":[{"b":0.000,"d":0.320,')
    assert '"t":"de(2)"' in line and '"t":"mètres(4)"' in line
    aset.free()


def test_many_filler_segments_for_a_short_text(oracle_mod, gpu_en, orc_en):
    """One word over audio full of alternating fillers: the backtrace has far more segments than
    a buffer sized from the word count (8 x words + 32).  ssw_first_pass_batch then reports how
    much room it needs (-(2 + k), not the 'no path' code -1), ssw_forced_align_batch searches
    again with that room, and the result is the reference's segmentation -- alone in its batch
    or next to a long text (the same utterance must not succeed in one batch and
    fail in another)."""
    lex = _lex(gpu_en, "en-us")
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    words = ["go"]
    nodes, _ = lex.first_pass_graph(words)
    name = [lex.word(int(n["wid"])) if (n["flags"] & 2) else None for n in nodes]

    def filler(state, w):
        for i, n in enumerate(nodes):
            if (n["flags"] & 2) and (n["flags"] & 1) and n["state"] == state and name[i] == w:
                return i
        raise AssertionError((state, w))

    kids = {}
    for i, n in enumerate(nodes):
        if n["parent"] >= 0:
            kids.setdefault(int(n["parent"]), []).append(i)
    path = []
    for k in range(90):
        path.append(filler(0, "<sil>" if k % 2 == 0 else "[NOISE]"))
    i = [i for i, n in enumerate(nodes) if (n["flags"] & 1) and n["state"] == 0
         and name[i] is None or (n["flags"] & 1) and n["state"] == 0 and name[i] == "go"][0]
    while True:
        path.append(i)
        if nodes[i]["flags"] & 2:
            break
        i = kids[i][0]
    for k in range(90):
        path.append(filler(1, "[NOISE]" if k % 2 == 0 else "<sil>"))
    rng = np.random.default_rng(11)
    rows = []
    for i in path:
        for sen in nodes[i]["senid"]:
            for _ in range(2):
                row = rng.integers(150, 400, gpu_en.n_sen).astype(np.int16)
                row[sen] = rng.integers(0, 20)
                rows.append(row)
    scr = np.stack(rows)
    want = F.first_pass(orc_en, olex, words, scr)
    assert want is not None and len(want) > 8 * len(words) + 32
    d = torch.from_numpy(scr).cuda()
    off = np.array([0, len(scr)], np.int32)
    # the plain C call with too little room says how much it needs
    n_seg, _ = lex.first_pass_raw(d, off, [words], max_seg=40)
    assert n_seg[0] == -(2 + len(want))
    n_seg, seg = lex.first_pass_raw(d, off, [words], max_seg=len(want))
    assert n_seg[0] == len(want)
    # decoder_alignment for a batch: alone, and beside a 30-word text
    alone = ssw.forced_alignment(gpu_en, lex, d, off, [words])[0]
    assert alone is not None
    assert [(w, int(a[0]), int(a[0] + a[1] - 1)) for w, a in zip(alone["words"], alone["word_al"])] \
        == [(w, s, e) for (w, s, e, _) in want]
    long_text = ["go", "forward", "ten", "meters"] * 8
    d2 = torch.from_numpy(np.concatenate([scr, scr])).cuda()
    both = ssw.forced_alignment(gpu_en, lex, d2, np.array([0, len(scr), 2 * len(scr)], np.int32),
                                [words, long_text])
    assert both[0] is not None
    assert both[0]["words"] == alone["words"]
    assert np.array_equal(both[0]["state_al"], alone["state_al"])


def _forced_big(monkeypatch):
    monkeypatch.setenv("SSW_FP_KERNEL", "big")


def test_hbm_resident_kernel_equals_the_register_one(oracle_mod, gpu_en, gpu_fr, orc_en, orc_fr,
                                                     monkeypatch):
    """first_pass_big_kernel (node state in HBM, for texts beyond 4096 phone-tree HMMs) forced
    on small problems (SSW_FP_KERNEL=big): noisy en-us paths with near-ties and failures, and the
    fr-fr alternates that are pronounced alike (the twins' list-order bookkeeping); equal to the
    register kernel and to the oracle."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(7, 24 * 10)
    texts, scores = [], []
    for t in range(24):
        n = 1 + int(u[t * 10] * 8)
        texts.append([vocab[int(x * len(vocab))] for x in u[t * 10 + 1:t * 10 + 1 + n]])
        scores.append(synth_scores(F, orc_en, olex, texts[-1], 100 + t, orc_en.n_sen,
                                   noise_lo=120 if t % 4 else 25))
    small = _first_pass(gpu_en, lex, scores, texts)
    Ff, olexf = _olex(oracle_mod, orc_fr, "fr-fr")
    lexf = _lex(gpu_fr, "fr-fr")
    textsf, scoresf = [], []
    for t in range(24):
        words = [["abus", "ait", "mauritaniens"], ["ait", "abus"], ["mauritaniens", "abus", "abus"],
                 ["abus"]][t % 4]
        textsf.append(words)
        scoresf.append(synth_scores(Ff, orc_fr, olexf, words, 300 + t, orc_fr.n_sen, sil_p=0.5))
    smallf = _first_pass(gpu_fr, lexf, scoresf, textsf)
    _forced_big(monkeypatch)
    # round 3: forced on small problems the long-text path starts with first_pass_win_kernel
    # (HMMs in registers, a sliding window of nodes); windows of 256, 512 and 1024 nodes (the
    # smallest slides and sometimes gives up: the call then falls back), then without it
    # (SSW_FP_WIN=0: first_pass_big_kernel, node state in HBM)
    for tpb in ("256", "512", "1024"):
        monkeypatch.setenv("SSW_FP_WIN_TPB", tpb)
        assert _first_pass(gpu_en, lex, scores, texts) == small, tpb
        assert _first_pass(gpu_fr, lexf, scoresf, textsf) == smallf, tpb
    monkeypatch.delenv("SSW_FP_WIN_TPB")
    monkeypatch.setenv("SSW_FP_HIST_BAND", "1")     # the window kernel's history budget overflows
    assert _first_pass(gpu_en, lex, scores, texts) == small
    monkeypatch.delenv("SSW_FP_HIST_BAND")
    monkeypatch.setenv("SSW_FP_WIN", "0")
    big = _first_pass(gpu_en, lex, scores, texts)
    bigf = _first_pass(gpu_fr, lexf, scoresf, textsf)
    assert big == small and bigf == smallf
    assert sum(g is not None for g in big) >= 12 and all(g is not None for g in bigf)
    # round 3: the kernel walks a band of grammar states and keeps a banded history table; without
    # the band (SSW_FP_BAND=0) and with a history budget so small that it overflows and the batch
    # falls back to the full table (SSW_FP_HIST_BAND=1) the result is the same
    monkeypatch.setenv("SSW_FP_BAND", "0")
    assert _first_pass(gpu_en, lex, scores, texts) == small
    assert _first_pass(gpu_fr, lexf, scoresf, textsf) == smallf
    monkeypatch.delenv("SSW_FP_BAND")
    monkeypatch.setenv("SSW_FP_HIST_BAND", "1")
    assert _first_pass(gpu_en, lex, scores, texts) == small
    assert _first_pass(gpu_fr, lexf, scoresf, textsf) == smallf
    monkeypatch.delenv("SSW_FP_HIST_BAND")
    for t in range(0, 24, 5):
        want = F.first_pass(orc_en, olex, texts[t], scores[t])
        assert (want is None and big[t] is None) or \
            [(w, s, s + d - 1, sc) for (w, s, d, sc) in big[t]] == want


def test_a_level_reruns_only_what_it_left_unfinished(oracle_mod, gpu_en, orc_en, monkeypatch):
    """when a level of the long-text path gives up on SOME utterances of a batch, only
    those are searched again one level down (FirstPassParams::only); the others keep what they
    have.  One-word texts (a handful of word-final HMMs: within a history budget of 24 entries
    per frame) beside eight-word texts (beyond it), interleaved, with the long-text kernels
    forced on them: every segmentation equals the unforced call's, whichever level wrote it."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(23, 20 * 10)
    texts, scores = [], []
    for t in range(20):
        n = 1 if t % 2 == 0 else 8
        texts.append([vocab[int(x * len(vocab))] for x in u[t * 10 + 1:t * 10 + 1 + n]])
        scores.append(synth_scores(F, orc_en, olex, texts[-1], 500 + t, orc_en.n_sen))
    plain = _first_pass(gpu_en, lex, scores, texts)
    assert sum(g is not None for g in plain) >= 16
    _forced_big(monkeypatch)
    for budget in ("24", "6", "1"):
        monkeypatch.setenv("SSW_FP_HIST_BAND", budget)
        assert _first_pass(gpu_en, lex, scores, texts) == plain, budget
        monkeypatch.setenv("SSW_FP_WIN", "0")          # ... and from the HBM-resident level down
        assert _first_pass(gpu_en, lex, scores, texts) == plain, budget
        monkeypatch.delenv("SSW_FP_WIN")
    for t in (0, 1, 6, 7):
        want = F.first_pass(orc_en, olex, texts[t], scores[t])
        assert (want is None and plain[t] is None) or \
            [(w, s, s + d - 1, sc) for (w, s, d, sc) in plain[t]] == want


@pytest.mark.timeout(900)
def test_a_page_of_2000_words(oracle_mod, gpu_en, orc_en):
    """a 2,000-word text (about 19 K phone-tree HMMs, 8 K word-final ones,
    ~90 K frames) through the HBM-resident first pass: the word segmentation of the oracle's
    fsg_search, word for word and frame for frame."""
    import time
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    words = [vocab[int(x * len(vocab))] for x in lcg_uniform(11, 2000)]
    assert len(lex.first_pass_graph(words, max_nodes=1 << 17)[0]) > 4 * 4096
    scr = synth_scores(F, orc_en, olex, words, 5, orc_en.n_sen, sil_p=0.1)
    t0 = time.time()
    got = _first_pass(gpu_en, lex, [scr], [words])[0]
    t1 = time.time()
    # (round 3: that was the sliding-window register kernel; the HBM-resident one must agree)
    import os
    os.environ["SSW_FP_WIN"] = "0"
    try:
        assert _first_pass(gpu_en, lex, [scr], [words])[0] == got
    finally:
        del os.environ["SSW_FP_WIN"]
    t1b = time.time()
    want = F.first_pass(orc_en, olex, words, scr)
    print(f"first pass with the node state in HBM: {t1b - t1:.2f} s")
    print(f"2000 words, {len(scr)} frames: GPU first pass {t1 - t0:.2f} s, oracle {time.time() - t1b:.2f} s")
    assert want is not None and got is not None and len(got) >= 2000
    assert [(w, s, s + d - 1, x) for (w, s, d, x) in got] == want
    # ... and the whole decoder_alignment: populate with those windows, the second pass over
    # ~9,000 phones (HBM-resident state, a 20 GB token table), propagate; against the oracle's
    # state_align_search on the same phones and windows
    off = np.array([0, len(scr)], np.int32)
    d = torch.from_numpy(np.ascontiguousarray(scr, np.int16)).cuda()
    t2 = time.time()
    aset = ssw.forced_align_batch(gpu_en, lex, d, off, [words])
    t3 = time.time()
    assert aset.status(0) == 0
    a = aset.utterance(0)
    aset.free()
    assert a["words"] == [w for (w, _, _, _) in got] and len(a["cipid"]) > 2560
    assert [tuple(int(x) for x in r[:2]) for r in a["word_al"]] == [(s, dd) for (_, s, dd, _) in got]
    pop = lex.populate([w for (w, _, _, _) in got], [s for (_, s, _, _) in got],
                       [dd for (_, _, dd, _) in got])
    st, du = pop["start"], pop["duration"]
    sf = np.maximum(st, 0).astype(np.int32)
    ef = np.where(du > 0, st + du, 2**31 - 1).astype(np.int32)
    senid = orc_en.sseq[pop["ssid"]].astype(np.uint16)
    init = np.zeros((len(st) * 3, 3), np.int32)      # what alignment_populate leaves in the states
    init[:, 0] = np.repeat(st, 3)
    init[:, 1] = np.repeat(du, 3)
    rv, rst, _ = orc_en.state_align(scr, senid, pop["tmatid"].astype(np.int16), sf=sf, ef=ef,
                                    state_init=init)
    print(f"decoder_alignment of the page: {t3 - t2:.2f} s, {len(a['cipid'])} phones; oracle second pass {time.time() - t3:.2f} s")
    assert rv == 0 and np.array_equal(a["state_al"], rst)


@pytest.mark.timeout(600)
def test_long_and_short_texts_in_one_batch(oracle_mod, gpu_en, orc_en):
    """A 700-word text (beyond the register kernel's 4096 HMMs and the LDS alignment kernel's
    2,560 phones) between two short ones in ONE call: the batch goes through the HBM-resident
    kernels as a whole, and every utterance must come out as when it is aligned alone."""
    F, olex = _olex(oracle_mod, orc_en, "en-us")
    lex = _lex(gpu_en, "en-us")
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    u = lcg_uniform(91, 720)
    texts = [[vocab[int(x * len(vocab))] for x in u[:5]],
             [vocab[int(x * len(vocab))] for x in u[10:710]],
             [vocab[int(x * len(vocab))] for x in u[712:720]]]
    assert len(lex.first_pass_graph(texts[1], max_nodes=1 << 17)[0]) > 4096
    scores = [synth_scores(F, orc_en, olex, t, 70 + i, orc_en.n_sen, sil_p=0.1)
              for i, t in enumerate(texts)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in scores])]).astype(np.int32)
    d = torch.from_numpy(np.ascontiguousarray(np.concatenate(scores), np.int16)).cuda()
    both = ssw.forced_align_batch(gpu_en, lex, d, off, texts)
    for i in range(3):
        assert both.status(i) == 0, i
        a = both.utterance(i)
        di = torch.from_numpy(np.ascontiguousarray(scores[i], np.int16)).cuda()
        alone = ssw.forced_align_batch(gpu_en, lex, di, np.array([0, len(scores[i])], np.int32),
                                       [texts[i]])
        b = alone.utterance(0)
        alone.free()
        assert a["words"] == b["words"] and np.array_equal(a["state_al"], b["state_al"]), i
        want = F.first_pass(orc_en, olex, texts[i], scores[i])
        assert [w for (w, _, _, _) in want] == a["words"]
    assert len(both.utterance(1)["cipid"]) > 2560
    both.free()


@pytest.mark.timeout(600)
def test_sliding_window_with_twins(oracle_mod, gpu_fr, orc_fr, monkeypatch):
    """The window kernel's rings under the twins' bookkeeping: fr-fr texts of 130 words full of
    alternates that are pronounced alike (their word-final HMMs follow the reference's list order
    through per-group records of node ids and the nodes' flags, which live in a ring here), long
    enough that a window of 256 nodes slides dozens of times.  Equal to the register kernel
    (two nodes per thread) and, for one of them, to the oracle."""
    F, olex = _olex(oracle_mod, orc_fr, "fr-fr")
    lex = _lex(gpu_fr, "fr-fr")
    pool = ["abus", "ait", "mauritaniens", "avance", "de", "dix", "mètres"]
    pool = [w for w in pool if w in olex.pron]
    assert len(pool) >= 4
    u = lcg_uniform(4242, 3 * 130)
    texts = [[pool[int(x * len(pool))] for x in u[t * 130:(t + 1) * 130]] for t in range(3)]
    scores = [synth_scores(F, orc_fr, olex, t, 800 + i, orc_fr.n_sen, sil_p=0.3,
                           noise_lo=60 if i else 120)
              for i, t in enumerate(texts)]
    n_nodes = [len(lex.first_pass_graph(t, max_nodes=1 << 16)[0]) for t in texts]
    assert min(n_nodes) > 1024 and max(n_nodes) <= 2048, n_nodes
    small = _first_pass(gpu_fr, lex, scores, texts)       # first_pass_kernel<2, 1024>
    assert all(g is not None for g in small)
    monkeypatch.setenv("SSW_FP_KERNEL", "big")
    for tpb in ("256", "512"):
        monkeypatch.setenv("SSW_FP_WIN_TPB", tpb)
        assert _first_pass(gpu_fr, lex, scores, texts) == small, tpb
    want = F.first_pass(orc_fr, olex, texts[0], scores[0])
    assert [(w, s, s + d - 1, x) for (w, s, d, x) in small[0]] == want
