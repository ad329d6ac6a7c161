"""Reference outputs measured by the round-4 judge with the REAL library ("(c)":
/root/reference configured out of tree, CMake Release, -ffp-contract=off, driven through
decoder_set_align_text -> decoder_alignment), held here as constants: 7 en-us texts over
tests/data/goforward.raw (first-pass words with frames AND every phone `start+dur(score)` of
decoder_alignment, compallsen=yes), 5 fr-fr texts over goforward_fr.raw (first-pass words with
frames) -- multi-filler paths, alternates, wrong word order, a text without a path.  The oracle
pipeline (front end -> features -> PTM -> fsg_oracle.first_pass -> populate -> second pass after
the rewind with carried history -> state_align) must reproduce every one of them; the GPU
counterparts are in tests/test_gpu_reference_pins.py.

Also here: the reference's own golden table tests/_test_feat.res lines 13-18 (tests/test_feat.c:
93-104, `1s_c_d_dd` with cmn=none), committed as tests/golden/test_feat_1s_c_d_dd.res.
"""
import os

import numpy as np
import pytest

from tests.conftest import MODEL_ROOT, ROOT
from tests.test_oracle_e2e_goforward import goforward_features, populate

# text -> (first pass "word sf ef" (inclusive), phones "NAME start+dur(score)"); None = no path
REF_EN_TEXTS = {
    "go forward": (
        "<sil> 0 45 go 46 63 forward 64 128 <sil> 129 212 <sil> 213 277",
        "SIL 0+46(-90) G 46+8(-183) OW 54+10(-202) F 64+14(-286) AO 78+6(-204) R 84+10(-161) "
        "W 94+7(-124) ER 101+11(-206) D 112+17(-825) SIL 129+84(-10216) SIL 213+65(-724)"),
    "go ten meters forward": (
        "<sil> 0 45 go 46 64 ten 65 87 meters 88 120 <sil> 121 163 forward 164 208 <sil> 209 277",
        "SIL 0+46(-90) G 46+8(-183) OW 54+11(-462) T 65+12(-831) EH 77+7(-1222) N 84+4(-670) "
        "M 88+3(-580) IY 91+3(-663) T 94+3(-468) ER 97+16(-1709) Z 113+8(-420) SIL 121+43(-3965) "
        "F 164+3(-652) AO 167+3(-724) R 170+3(-752) W 173+3(-586) ER 176+13(-617) D 189+20(-1691) "
        "SIL 209+69(-1084)"),
    "forward ten": (
        "<sil> 0 60 forward 61 116 ten 117 156 <sil> 157 212 <sil> 213 277",
        "SIL 0+61(-2015) F 61+17(-752) AO 78+6(-204) R 84+10(-161) W 94+7(-124) ER 101+11(-206) "
        "D 112+5(-81) T 117+15(-588) EH 132+9(-150) N 141+16(-662) SIL 157+56(-7740) "
        "SIL 213+65(-724)"),
    "hello world": (
        "<sil> 0 46 hello(2) 47 63 world 64 128 <sil> 129 212 <sil> 213 277",
        "SIL 0+47(-132) HH 47+5(-460) EH 52+3(-304) L 55+3(-333) OW 58+6(-368) W 64+17(-1730) "
        "ER 81+28(-1776) L 109+3(-712) D 112+17(-1122) SIL 129+84(-10216) SIL 213+65(-724)"),
    "ten": (
        "<sil> 0 109 ten 110 156 <sil> 157 212 <sil> 213 277",
        "SIL 0+110(-8939) T 110+22(-1450) EH 132+9(-150) N 141+16(-662) SIL 157+56(-7740) "
        "SIL 213+65(-724)"),
    # (SURVEY Appendix C's recording of the same text, tests/test_oracle_e2e_goforward.py)
    "go forward ten meters": (
        "<sil> 0 45 go 46 63 forward 64 116 ten 117 152 meters 153 210 <sil> 211 277",
        "SIL 0+46(-90) G 46+8(-183) OW 54+10(-202) F 64+14(-286) AO 78+6(-204) R 84+10(-161) "
        "W 94+7(-124) ER 101+11(-206) D 112+5(-81) T 117+15(-588) EH 132+9(-150) N 141+12(-439) "
        "M 153+6(-72) IY 159+12(-580) T 171+3(-250) ER 174+16(-281) Z 190+21(-626) SIL 211+67(-900)"),
    "go go forward ten meters meters": None,      # decoder_alignment returns NULL
}
REF_FR_TEXTS = {
    "avance": "<sil> 0 31 avance(2) 32 142 <sil> 143 238",
    "dix mètres avance de": "dix 0 5 <sil> 6 32 mètres 33 50 avance(2) 51 99 de(2) 100 142 <sil> 143 238",
    "bonjour le monde": "<sil> 0 28 bonjour 29 71 <sil> 72 84 le 85 142 <sil> 143 221 monde(2) 222 238",
    "de de de": "<sil> 0 31 de(2) 32 69 <sil> 70 84 de(2) 85 99 de(2) 100 142 <sil> 143 238",
    # SURVEY Appendix C (tests/test_first_pass_oracle.py)
    "avance de dix mètres": "<sil> 0 31 avance 32 79 de(2) 80 99 dix 100 117 mètres(4) 118 166 <sil> 167 238",
}


def parse_words(s):
    t = s.split()
    return [(t[i], int(t[i + 1]), int(t[i + 2])) for i in range(0, len(t), 3)]


def parse_phones(s):
    out = []
    t = s.split()
    for name, rest in zip(t[0::2], t[1::2]):
        se, sc = rest.split("(")
        a, d = se.split("+")
        out.append((name, int(a), int(d), int(sc.rstrip(")"))))
    return out


def alignment_inputs(O, m, words):
    """words [(name, start, dur)] -> what state_align_search_init reads
    (src/state_align_search.c:458-471) after alignment_populate"""
    phones = populate(O, m, words)
    senid = m.sseq[np.array([p[1] for p in phones])]
    tmat = np.array([p[2] for p in phones], np.int16)
    wstart = np.array([words[p[3]][1] for p in phones], np.int32)
    wdur = np.array([words[p[3]][2] for p in phones], np.int32)
    sf = np.where(wstart > 0, wstart, 0).astype(np.int32)
    ef = np.where(wdur > 0, wstart + wdur, 2**31 - 1).astype(np.int32)
    state_init = np.stack([np.repeat(wstart, 3), np.repeat(wdur, 3),
                           np.zeros(3 * len(phones), np.int32)], 1).astype(np.int32)
    return phones, senid, tmat, sf, ef, state_init


@pytest.fixture(scope="module")
def en(oracle_mod):
    """features of goforward.raw, first-pass scores (history carried frame to frame from the
    reset state) and second-pass scores (again from frame 0 after acmod_rewind, history carried
    over: src/decoder.c:786-793) -- the same for every text"""
    from oracle import fsg_oracle as F
    O = oracle_mod
    d = os.path.join(MODEL_ROOT, "en-us")
    m = O.Model(d)
    lex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    feats = goforward_features(O)
    m.ptm_reset()
    first = m.ptm_score_utt(feats)
    m.ptm_set_frame_idx(0)
    second = np.zeros_like(first)
    for t in range(len(feats)):
        second[t] = m.ptm_frame_eval(feats[t], t)
        m.ptm_set_frame_idx(t + 1)
    return O, m, lex, first, second


@pytest.mark.parametrize("text", list(REF_EN_TEXTS))
def test_en_us_text_reproduces_the_real_library(en, text):
    from oracle import fsg_oracle as F
    O, m, lex, first, second = en
    seg = F.first_pass(m, lex, text.split(), first)
    want = REF_EN_TEXTS[text]
    if want is None:
        assert seg is None
        return
    assert seg is not None
    assert [(w, sf, ef) for (w, sf, ef, _) in seg] == parse_words(want[0])
    words = [(w, sf, ef - sf + 1) for (w, sf, ef, _) in seg]
    phones, senid, tmat, sf, ef, state_init = alignment_inputs(O, m, words)
    rv, st, ph = m.state_align(second, senid, tmat, sf=sf, ef=ef, state_init=state_init)
    assert rv == 0
    got = [(phones[i][0], int(ph[i, 0]), int(ph[i, 1]), int(ph[i, 2])) for i in range(len(ph))]
    assert got == parse_phones(want[1])


@pytest.fixture(scope="module")
def fr(oracle_mod):
    from oracle import fsg_oracle as F
    from tests.test_first_pass_oracle import features
    O = oracle_mod
    d = os.path.join(MODEL_ROOT, "fr-fr")
    m = O.Model(d)
    lex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    return m, lex, m.ptm_score_utt(features(O, "goforward_fr.raw"))


@pytest.mark.parametrize("text", list(REF_FR_TEXTS))
def test_fr_fr_first_pass_reproduces_the_real_library(fr, text):
    from oracle import fsg_oracle as F
    m, lex, scr = fr
    seg = F.first_pass(m, lex, text.split(), scr)
    assert seg is not None
    assert [(w, sf, ef) for (w, sf, ef, _) in seg] == parse_words(REF_FR_TEXTS[text])


def test_dynamic_features_match_the_reference_golden_table(oracle_mod):
    """tests/_test_feat.res lines 13-18 (tests/test_feat.c:93-104: feat=1s_c_d_dd, cmn=none,
    the six 13-dim rows of tests/test_feat.c:14-39, printed %.3f): all 39 columns with cmn off;
    with the batch CMN the path uses, the delta and delta-delta columns (a per-column constant
    cancels in a difference)."""
    gold = np.loadtxt(os.path.join(ROOT, "tests", "golden", "test_feat_1s_c_d_dd.res"))
    assert gold.shape == (6, 39)
    cep = gold[:, :13].astype(np.float32)          # the statics ARE the input rows
    out = oracle_mod.feat_1s_c_d_dd(cep, cmn=False)
    assert np.abs(out - gold).max() < 1e-3         # (the table is printed to three decimals)
    with_cmn = oracle_mod.feat_1s_c_d_dd(cep)
    assert np.abs(with_cmn[:, 13:] - gold[:, 13:]).max() < 1e-3
    assert np.abs(with_cmn[:, :13] + cep.mean(0) - gold[:, :13]).max() < 1e-3
