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


# Appendix C, decoder_result_json(d, 0.0, 1) of the real library on goforward.wav (compallsen=yes):
# the part of the line the survey recorded (the hot path's output as the reference prints it)
REF_JSON_PREFIX = (
    '{"b":0.000,"d":2.790,"p":1.000,"t":"go forward ten meters","w":['
    '{"b":0.000,"d":0.460,"p":0.991,"t":"<sil>","w":[{"b":0.000,"d":0.460,"p":0.991,"t":"SIL"}]},'
    '{"b":0.460,"d":0.180,"p":0.962,"t":"go","w":[{"b":0.460,"d":0.080,"p":0.982,"t":"G"},'
    '{"b":0.540,"d":0.100,"p":0.980,"t":"OW"}]},'
    '{"b":0.640,"d":0.530,"p":0.899,"t":"forward","w":[{"b":0.640,"d":0.140,"p":0.972,"t":"F"},'
    '{"b":0.780,"d":0.060,"p":0.980,"t":"AO"},{"b":0.840,"d":0.100,"p":0.984,"t":"R"},'
    '{"b":0.940,"d":0.070,"p":0.988,"t":"W"},{"b":1.010,"d":0.110,"p":0.980,"t":"ER"},'
    '{"b":1.120,"d":0.050,"p":0.992,"t":"D"}]},')

REF_PHONES = ("SIL 0+46(-90) G 46+8(-183) OW 54+10(-202) F 64+14(-286) AO 78+6(-204) R 84+10(-161) "
              "W 94+7(-124) ER 101+11(-206) D 112+5(-81) T 117+15(-588) EH 132+9(-150) "
              "N 141+12(-439) M 153+6(-72) IY 159+12(-580) T 171+3(-250) ER 174+16(-281) "
              "Z 190+21(-626) SIL 211+67(-900)")


def test_alignment_json_reproduces_the_reference_line(lex_en):
    """ssw_alignment_json = decoder_result_json at align_level 1: fed with the reference's own
    recorded alignment (Appendix C) it must print the reference's own line."""
    import re

    model, lex = lex_en
    words = ["<sil>", "go", "forward", "ten", "meters", "<sil>"]
    rows = lex.populate(words)
    ph = [(int(a), int(b), int(c)) for a, b, c in re.findall(r"(\d+)\+(\d+)\((-?\d+)\)", REF_PHONES)]
    assert len(ph) == len(rows["cipid"]) == 18
    phone_al = np.array(ph, np.int32)
    word_al = model.propagate(phone_al, rows["parent"], len(words))
    assert word_al[:, 2].tolist() == [-90, -385, -1062, -1177, -1809, -900]
    line = lex.alignment_json("go forward ten meters", words, word_al, rows["cipid"],
                              rows["parent"], phone_al, n_frames=279)
    assert line.startswith(REF_JSON_PREFIX), line[:len(REF_JSON_PREFIX)]
    assert line.endswith(',{"b":2.110,"d":0.670,"p":0.914,"t":"<sil>","w":['
                         '{"b":2.110,"d":0.670,"p":0.914,"t":"SIL"}]}]}\n')
    import json
    doc = json.loads(line)
    assert [w["t"] for w in doc["w"]] == words and len(doc["w"][2]["w"]) == 6
    # align_level 2 adds the state level, named by senone id
    states = np.repeat(phone_al, 3, axis=0)
    senid = np.arange(54, dtype=np.uint16)
    line2 = lex.alignment_json("go forward ten meters", words, word_al, rows["cipid"],
                               rows["parent"], phone_al, n_frames=279, state_senid=senid,
                               state_al=states)
    doc2 = json.loads(line2)
    assert [s["t"] for s in doc2["w"][1]["w"][0]["w"]] == ["3", "4", "5"]
