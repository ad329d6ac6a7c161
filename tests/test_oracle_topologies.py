"""CPU: the oracle's restatements of hmm_vit_eval_5st_lr (src/hmm.c:166-304) and
hmm_vit_eval_anytopo (:671-739) against a second, independent transcription in Python, and the
loaders (oracle's and the library's host side) on a 5-state model made on the spot.  No reference
output exists for such HMMs (no shipped model has them): this is what pins the oracle's two
functions that tests/test_gpu_topologies.py then holds the GPU kernel to."""
import os

import numpy as np
import pytest

W = -536870912


def _anytopo(ne, tp, senscr, senid, score, hist, out_score, out_hist):
    """Every target from the exit state down takes the best existing arc into it: candidates in
    the order self-loop, nearest lower state, ..., state 0; a later candidate wins only when
    strictly better; without a winner among the lower states the history stays."""
    T = lambda i, j: -int(tp[i][j])
    st = [int(score[k]) - int(senscr[senid[k]]) for k in range(ne)]
    st = [st[0]] + [max(v, W) for v in st[1:]]
    new_s, new_h = list(map(int, score)), list(map(int, hist))
    best = None
    for to in range(ne, -1, -1):
        cur, frm = W, -1
        if to < ne and T(to, to) > -255:
            cur = st[to] + T(to, to)
        for f in range(to - 1, -1, -1):
            if T(f, to) > -255 and st[f] + T(f, to) > cur:
                cur, frm = st[f] + T(f, to), f
        if to == ne:
            out_score = cur
            if frm >= 0:
                out_hist = int(hist[frm])
        else:
            new_s[to] = cur
            if frm >= 0:
                new_h[to] = int(hist[frm])
        best = cur if best is None or cur > best else best
    return best, new_s, new_h, out_score, out_hist


def _pick3(t0, t1, t2):
    """0, 1 or 2: the reference's nested comparison (ties: t1 over t0, the winner over t2)"""
    w = 0 if t0 > t1 else 1
    return 2 if t2 > (t0, t1)[w] else w


def _five(tp, senscr, senid, score, hist, out_score, out_hist):
    T = lambda i, j: -int(tp[i][j])
    a = [int(score[k]) - int(senscr[senid[k]]) for k in range(5)]   # score + senone score
    s, h = list(map(int, score)), list(map(int, hist))
    best = W
    if a[3] > W:
        t1, t2 = a[4] + T(4, 5), a[3] + T(3, 5)
        out_score, out_hist = (t1, h[4]) if t1 > t2 else (t2, h[3])
        out_score = max(out_score, W)
        best = out_score
    loc = list(a)            # the local values the later blocks go on with
    if a[2] > W:
        c = (loc[4] + T(4, 4), loc[3] + T(3, 4), loc[2] + T(2, 4))
        k = _pick3(*c)
        h[4] = (h[4], h[3], h[2])[k]
        loc[4] = max(c[k], W)
        s[4] = loc[4]
        best = max(best, loc[4])
    if a[1] > W:
        c = (loc[3] + T(3, 3), loc[2] + T(2, 3), loc[1] + T(1, 3))
        k = _pick3(*c)
        h[3] = (h[3], h[2], h[1])[k]
        loc[3] = max(c[k], W)
        s[3] = loc[3]
        best = max(best, loc[3])
    c = (loc[2] + T(2, 2), loc[1] + T(1, 2), loc[0] + T(0, 2))
    k = _pick3(*c)
    h[2] = (h[2], h[1], h[0])[k]
    s[2] = max(c[k], W)
    t0, t1 = loc[1] + T(1, 1), loc[0] + T(0, 1)
    if not t0 > t1:
        h[1] = h[0]
    s[1] = max(t0 if t0 > t1 else t1, W)
    s[0] = max(loc[0] + T(0, 0), W)
    best = max(best, s[2], s[1], s[0])
    return best, s, h, out_score, out_hist


def _random_tp(rng, ne):
    tp = np.full((ne, ne + 1), 255, np.uint8)
    for j in range(ne):
        if rng.random() < 0.85 or ne == 5:
            tp[j, j] = rng.integers(0, 40)
        tp[j, j + 1] = rng.integers(0, 40)
        for d in (2, 3):
            if j + d <= ne and rng.random() < 0.4 and (ne != 5 or d == 2):
                tp[j, j + d] = rng.integers(0, 80)
    return tp


@pytest.mark.parametrize("ne", [1, 2, 4, 5])
def test_hmm_vit_eval_other_topologies_against_transcription(oracle_mod, ne):
    rng = np.random.default_rng(ne)
    for trial in range(1500):
        tp = _random_tp(rng, ne)
        senscr = rng.integers(0, 4000, 16).astype(np.int16)
        senid = rng.integers(0, 16, ne).astype(np.uint16)
        score = np.where(rng.random(ne) < 0.3, W, -rng.integers(0, 100000, ne)).astype(np.int32)
        if trial % 7 == 0:
            score[:] = W                    # an HMM nothing has entered yet
        if trial % 11 == 0:
            score[rng.integers(0, ne)] = W + rng.integers(0, 3000)   # just above the floor
        hist = rng.integers(-1, 50, ne).astype(np.int32)
        os0, oh0 = (W, -1) if trial % 3 else (-int(rng.integers(0, 9999)), int(rng.integers(0, 50)))
        got = oracle_mod.hmm_vit_eval(tp, senscr, senid, score, hist, os0, oh0)
        exp = (_five(tp, senscr, senid, score, hist, os0, oh0) if ne == 5
               else _anytopo(ne, tp, senscr, senid, score, hist, os0, oh0))
        assert got[0] == exp[0], (trial, got, exp)
        assert got[1].tolist() == exp[1] and got[2].tolist() == exp[2], (trial, got, exp)
        assert (got[3], got[4]) == (exp[3], exp[4]), (trial, got, exp)


def test_anytopo_agrees_with_3st_on_plain_left_to_right_matrices(oracle_mod, orc_en):
    """Two branches of one function: on en-us's own matrices (no skip arcs, every self-loop
    there) and live states, the generic evaluator and hmm_vit_eval_3st_lr give the same scores
    (their tie rules differ only where two arcs tie, and their floors only below WORST_SCORE)."""
    rng = np.random.default_rng(5)
    n = 0
    for trial in range(500):
        tp = orc_en.tp[rng.integers(0, 42)]
        senscr = rng.integers(0, 4000, 16).astype(np.int16)
        senid = rng.integers(0, 16, 3).astype(np.uint16)
        score = (-rng.integers(0, 100000, 3)).astype(np.int32)
        hist = rng.integers(0, 50, 3).astype(np.int32)
        a = oracle_mod.hmm_vit_eval(tp, senscr, senid, score, hist, W, -1)
        b = _anytopo(3, tp, senscr, senid, score, hist, W, -1)
        T = lambda i, j: -int(tp[i][j])
        s = [int(score[k]) - int(senscr[senid[k]]) for k in range(3)]
        ties = (s[2] + T(2, 3) == W or s[2] + T(2, 2) == s[1] + T(1, 2)
                or s[1] + T(1, 1) == s[0] + T(0, 1))
        if ties:
            continue
        n += 1
        assert a[0] == b[0] and a[1].tolist() == b[1] and a[3] == b[3]
        assert a[2].tolist() == b[2] and a[4] == b[4]
    assert n > 400


def test_loaders_agree_on_a_five_state_model(oracle_mod, tmp_path):
    import soundswallower_amd as ssw
    from soundswallower_amd import _lib
    from tests.test_gpu_topologies import _write_mdef, _write_tmat
    _lib.build()
    src = ssw.model_dir("en-us")
    mdef, tmat = str(tmp_path / "mdef5"), str(tmp_path / "tmat5")
    _write_mdef(mdef, 5, 105)
    _write_tmat(tmat, 42, 5, 205)
    kw = dict(mdef=mdef, means=os.path.join(src, "means"), sendump=os.path.join(src, "sendump"),
              tmat=tmat)
    g = ssw.Model(variances=os.path.join(src, "variances"), config={"device": -2}, **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    assert g.n_emit_state == 5 and g.tmat_n_emit == 5 and o.sseq.shape == (g.n_sseq, 5)
    assert o.tp.shape == (42, 5, 6)
    assert np.array_equal(g.table("tp"), o.tp.reshape(-1))
    assert np.array_equal(g.table("sseq"), o.sseq.reshape(-1))
    # the arcs the file holds came through tmat.c:206's arithmetic, the others are 255
    tp = o.tp
    assert (tp[:, np.arange(5), np.arange(5)] < 255).all() and (tp[:, 0, 3:] == 255).all()
    assert (tp[:, 0, 2] < 255).any() and (tp[:, 0, 2] == 255).any()
