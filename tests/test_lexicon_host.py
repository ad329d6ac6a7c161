"""CPU: the product's dictionary / triphone lookup / alignment_populate (host C) against the
oracle's independent restatement, and against the reference's recorded phone string."""
import os

import numpy as np
import pytest

import soundswallower_amd as ssw
from tests.conftest import MODEL_ROOT
from tests.test_oracle_e2e_goforward import REF_WORDS, _parse_ref, populate


@pytest.fixture(scope="module")
def lex_en():
    d = os.path.join(MODEL_ROOT, "en-us")
    m = ssw.Model(d, config={"device": -2})
    return m, ssw.Lexicon(m, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))


def test_dictionary_loads(lex_en):
    m, lex = lex_en
    assert len(lex) > 130000
    assert lex.pron("forward") == ["F", "AO", "R", "W", "ER", "D"]
    assert lex.pron("<sil>") == ["SIL"] and lex.pron("</s>") == ["SIL"]
    with pytest.raises(KeyError):
        lex.pron("xyzzyplugh")


def test_triphone_lookup_matches_oracle(lex_en, oracle_mod, orc_en):
    m, lex = lex_en
    rng = np.random.default_rng(12)
    n_ci = orc_en.n_ciphone
    for _ in range(4000):
        b, l, r = (int(x) for x in rng.integers(0, n_ci, 3))
        pos = int(rng.integers(0, 4))
        assert lex.phone_id_nearest(b, l, r, pos) == oracle_mod.phone_id_nearest(orc_en, b, l, r, pos)


def test_populate_goforward_is_the_reference_phone_string(lex_en, oracle_mod, orc_en):
    m, lex = lex_en
    words = [w for (w, _, _, _) in REF_WORDS]
    start = [s for (_, s, _, _) in REF_WORDS]
    dur = [d for (_, _, d, _) in REF_WORDS]
    got = lex.populate(words, start, dur)
    names = [lex._L.ssw_ciphone_name(m._m, int(c)).decode() for c in got["cipid"]]
    assert names == [r[0] for r in _parse_ref()]            # js/tests.js:127-130 plus the SILs
    ref = populate(oracle_mod, orc_en, list(zip(words, start, dur)))
    assert got["ssid"].tolist() == [p[1] for p in ref]
    assert got["tmatid"].tolist() == [p[2] for p in ref]
    assert got["parent"].tolist() == [p[3] for p in ref]
    assert got["start"].tolist() == [start[p[3]] for p in ref]


def test_populate_random_sentences_match_oracle(lex_en, oracle_mod, orc_en):
    m, lex = lex_en
    with open(os.path.join(MODEL_ROOT, "en-us", "dict.txt")) as fh:
        vocab = [ln.split()[0] for ln in fh if ln.strip()]
    rng = np.random.default_rng(3)
    for _ in range(20):
        words = ["<sil>"] + [vocab[i] for i in rng.integers(0, len(vocab), 12)] + ["<sil>"]
        got = lex.populate(words)
        ref = populate(oracle_mod, orc_en, [(w, 0, 0) for w in words])
        assert got["ssid"].tolist() == [p[1] for p in ref]
        assert got["tmatid"].tolist() == [p[2] for p in ref]
    with pytest.raises(ssw.SswError, match="not in the dictionary"):
        lex.populate(["go", "xyzzyplugh"])
