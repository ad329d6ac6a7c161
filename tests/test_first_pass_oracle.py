"""Pin of the first-pass restatement (oracle/fsg_oracle.py) against the reference's own output.

SURVEY.md Appendix C records the word segmentation the real library produced for its two test
recordings -- en-us "go forward ten meters" and fr-fr "avance de dix mètres" (where the first
pass had to pick alternate pronunciations, de(2) and mètres(4), and a one-phone word exists).
The first pass decides exactly that: which words / fillers / alternates, and their frames.

tests/golden/goforward.raw and goforward_fr.raw are the reference's tests/data recordings.
"""
import os

import numpy as np
import pytest

from tests.conftest import MODEL_ROOT, ROOT

REF_EN = [("<sil>", 0, 46), ("go", 46, 18), ("forward", 64, 53), ("ten", 117, 36),
          ("meters", 153, 58), ("<sil>", 211, 67)]
REF_FR = [("<sil>", 0, 32), ("avance", 32, 48), ("de(2)", 80, 20), ("dix", 100, 18),
          ("mètres(4)", 118, 49), ("<sil>", 167, 72)]


def features(O, raw):
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", raw), dtype="<i2")
    cep = O.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                    transform="dct")
    return O.feat_1s_c_d_dd(cep)


def run(O, name, raw, text):
    from oracle import fsg_oracle as F
    d = os.path.join(MODEL_ROOT, name)
    m = O.Model(d)
    lex = F.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
    feats = features(O, raw)
    senscr = m.ptm_score_utt(feats)
    trace = []
    seg = F.first_pass(m, lex, text.split(), senscr, trace=trace)
    return seg, trace, len(feats)


def test_first_pass_reproduces_the_reference_segmentation_en(oracle_mod):
    seg, trace, n = run(oracle_mod, "en-us", "goforward.raw", "go forward ten meters")
    assert seg is not None
    got = [(w, sf, ef - sf + 1) for (w, sf, ef, _) in seg]
    assert got == REF_EN and n == 278
    assert max(t[2] for t in trace) < 200          # the beam keeps the active set small


def test_first_pass_reproduces_the_reference_segmentation_fr(oracle_mod):
    seg, trace, n = run(oracle_mod, "fr-fr", "goforward_fr.raw", "avance de dix mètres")
    assert seg is not None
    got = [(w, sf, ef - sf + 1) for (w, sf, ef, _) in seg]
    assert got == REF_FR and n == 239
