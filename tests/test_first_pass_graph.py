"""The host-side graph builder of the first pass (csrc/ssw_fsg.c) against the oracle's lextree
(oracle/fsg_oracle.py, which follows src/fsg_lextree.c node by node): same HMMs, same entry
penalties, same predecessors, same context sets, same beams -- for the reference's two test
sentences, for texts with one-phone words, repeated words and alternates, and for random texts."""
import os

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import lcg_uniform
from tests.conftest import MODEL_ROOT


def canon_oracle(F, O, m, lex, words, cfg=None):
    cfg = cfg or F.Config
    lmath = O.Logmath(1.0001, 0)
    lw = np.float32(cfg.lw)
    pip = int(np.float32(lmath.log(cfg.pip)) * lw) >> 10
    wip = int(np.float32(lmath.log(cfg.wip)) * lw) >> 10
    arcs = F.build_fsg(lex, words, lmath, cfg)
    nodes, roots = F.build_lextree(m, lex, arcs, wip, pip)
    parent = {}
    state_of = {}
    for s, r in enumerate(roots):
        while r is not None:
            state_of[r.idx] = s
            r = r.sibling
    for n in nodes:
        c = n.succ
        while c is not None:
            parent.setdefault(c.idx, n.idx)      # word-initial nodes sharing a subtree: any one
            c = c.sibling
    def key(n):
        sen = tuple(int(x) for x in m.sseq[n.ssid])
        ctxt = n.ctxt & ((1 << 64) - 1) if (n.ppos == 0 or n.leaf) else 0
        allrc = n.leaf and (n.link.filler or len(lex.pron[n.link.word]) == 1)
        return (sen, n.tmat, n.logs2prob, n.ppos == 0, bool(n.leaf), bool(allrc), n.ci_ext,
                state_of.get(n.idx, -1), n.link.to if n.leaf else -1,
                n.link.word if n.leaf else None, ctxt)
    keys = {n.idx: key(n) for n in nodes}
    out = []
    for n in nodes:
        chain = []
        k = n.idx
        while k in parent:
            k = parent[k]
            chain.append(keys[k][:3])
        out.append((keys[n.idx], tuple(chain)))
    return sorted(out, key=repr)


def canon_product(lexp, m, words, cfg=None):
    nodes, beams = lexp.first_pass_graph(words, cfg=cfg)
    keys = []
    for n in nodes:
        fl = int(n["flags"])
        root, leaf, allrc = bool(fl & 1), bool(fl & 2), bool(fl & 4)
        ctxt = int(n["ctxt"]) if (root or leaf) else 0
        keys.append((tuple(int(x) for x in n["senid"]), int(n["tmat"]), int(n["pen"]), root, leaf,
                     allrc, int(n["ci_ext"]), int(n["state"]) if root else -1,
                     int(n["to_state"]) if leaf else -1,
                     lexp.word(int(n["wid"])) if leaf else None, ctxt))
    out = []
    for i, n in enumerate(nodes):
        chain = []
        k = i
        while nodes[k]["parent"] >= 0:
            k = int(nodes[k]["parent"])
            chain.append(keys[k][:3])
        out.append((keys[i], tuple(chain)))
    return sorted(out, key=repr), beams


@pytest.fixture(scope="module")
def both(oracle_mod):
    from oracle import fsg_oracle as F
    out = {}
    for name in ("en-us", "fr-fr"):
        d = os.path.join(MODEL_ROOT, name)
        mo = oracle_mod.Model(d)
        lo = F.Lexicon(mo, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
        mp = ssw.Model(d, config={"device": -2})
        lp = ssw.Lexicon(mp, os.path.join(d, "dict.txt"), os.path.join(d, "noisedict.txt"))
        out[name] = (F, mo, lo, mp, lp)
    return out


TEXTS = {
    "en-us": ["go forward ten meters", "a", "i a i", "the the the", "either way read the record",
              "forward"],
    "fr-fr": ["avance de dix mètres", "de", "à de à", "dix dix"],
}


@pytest.mark.parametrize("name", ["en-us", "fr-fr"])
def test_graph_matches_the_oracle_lextree(oracle_mod, both, name):
    F, mo, lo, mp, lp = both[name]
    lmath = oracle_mod.Logmath(1.0001, 0)
    for text in TEXTS[name]:
        words = text.split()
        want = canon_oracle(F, oracle_mod, mo, lo, words)
        got, beams = canon_product(lp, mp, words)
        assert len(got) == len(want), text
        assert got == want, text
        assert beams.tolist() == [int(lmath.log(F.Config.beam)) >> 10,
                                  int(lmath.log(F.Config.pbeam)) >> 10,
                                  int(lmath.log(F.Config.wbeam)) >> 10]


def test_graph_matches_on_random_texts(oracle_mod, both):
    F, mo, lo, mp, lp = both["en-us"]
    vocab = [w for w in lo.order[:lo.filler_start] if "(" not in w]
    u = lcg_uniform(99, 40 * 12)
    k = 0
    for t in range(40):
        n = 1 + int(u[k] * 11)
        words = [vocab[int(x * len(vocab))] for x in u[k + 1:k + 1 + n]]
        k += 12
        assert canon_product(lp, mp, words)[0] == canon_oracle(F, oracle_mod, mo, lo, words), words


def test_dictionary_alternates_and_fillers(both):
    F, mo, lo, mp, lp = both["en-us"]
    assert lp.word(lp.word_id("forward")) == "forward" and lp.word_id("nope-nope") == -1
    assert len(lp) == len(lo.order)
    # unknown words are refused with the reference's message
    with pytest.raises(ssw.SswError, match="Unknown word"):
        lp.first_pass_graph(["go", "qqqqq"])


def test_alternates_pronounced_alike_are_marked(both):
    """fr-fr lists alternates with identical pronunciations (abus / abus(2); one-phone ait /
    ait(2)): their word-final HMMs can never differ, and the kernel needs to know which of a
    group the reference's list order would let through (flags 8 member, 16 first, 32 last)."""
    F, mo, lo, mp, lp = both["fr-fr"]
    nodes, _ = lp.first_pass_graph(["abus", "ait"])
    twins = [n for n in nodes if n["flags"] & 8]
    names = sorted({lp.word(int(n["wid"])) for n in twins})
    assert names == ["abus", "abus(2)", "ait", "ait(2)"]
    for w in ("abus", "ait"):
        grp = [n for n in twins if lp.word(int(n["wid"])).split("(")[0] == w]
        by_sen = {}
        for n in grp:
            by_sen.setdefault(tuple(n["senid"]), []).append(n)
        for members in by_sen.values():
            assert len(members) == 2
            assert sorted(int(n["flags"]) & 48 for n in members) == [16, 32]
    # links in the reference's list order: alternates first, the word itself last
    words = [lp.word(int(n["wid"])) for n in nodes if n["flags"] & 2 and n["state"] == 0
             and not lp.word(int(n["wid"])).startswith("<")]
    assert words.index("abus(2)") < words.index("abus")
    en_nodes, _ = both["en-us"][4].first_pass_graph(["go", "forward"])
    assert not any(n["flags"] & 8 for n in en_nodes)


@pytest.mark.parametrize("kw", [dict(use_filler=0), dict(use_altpron=0),
                                dict(lw=9.5, wip=0.2, pip=0.5, silprob=0.1, fillprob=1e-3),
                                dict(use_filler=0, use_altpron=0)])
def test_graph_follows_the_configuration(oracle_mod, both, kw):
    """fsgusefiller / fsgusealtpron change which links exist, lw / wip / pip / silprob / fillprob
    the entry penalties: same graphs as the oracle under the same settings (fr-fr)."""
    F, mo, lo, mp, lp = both["fr-fr"]
    names = {"use_filler": "fsgusefiller", "use_altpron": "fsgusealtpron"}
    ocfg = type("Cfg", (F.Config,), {names.get(k, k): (bool(v) if k in names else v)
                                     for k, v in kw.items()})
    cfg = lp.first_pass_config(**kw)
    for text in TEXTS["fr-fr"]:
        words = text.split()
        assert canon_product(lp, mp, words, cfg)[0] == canon_oracle(F, oracle_mod, mo, lo, words, ocfg), (kw, text)
