"""HMMs that are not 3-state (SURVEY 8 rows a16 / a17's neighbours): hmm_vit_eval takes
hmm_vit_eval_5st_lr for 5 emitting states and hmm_vit_eval_anytopo for every other count up to
HMM_MAX_NSTATE = 5 (src/hmm.c:741-759, :166-304, :671-739).  Neither shipped model has such
HMMs, so the models here are made on the spot: en-us's mdef with its senone sequences stretched
or cut to n_emit states, and a random transition_matrices file of that shape (Bakis arcs, skip
arcs in some matrices, a state without a self-loop here and there for the generic evaluator).
GPU (viterbi_align_any_kernel through ssw_align_batch and the state_align_search object)
against the oracle's restatement of the same reference functions."""
import os
import struct

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import lcg_uniform, synth_alignment_task, synth_features

pytestmark = pytest.mark.gpu


def _write_mdef(dst, ne, seed):
    """en-us's BMDF with n_emit_state = ne: header word 2 and the senone-sequence block (the
    file's tail, uint16 [n_sseq][n_emit]) rewritten; names, tree and phone table as they are."""
    b = open(os.path.join(ssw.model_dir("en-us"), "mdef"), "rb").read()
    magic, ver, dl = struct.unpack("<3i", b[:12])
    assert magic == 0x46444D42
    at = 12 + dl
    hd = list(struct.unpack("<10i", b[at:at + 40]))
    hd_at = at
    at += 40
    names = at
    for _ in range(hd[0]):
        at = b.index(b"\0", at) + 1
    at = names + ((at - names + 3) // 4) * 4
    at += hd[8] * 8 + hd[1] * 12
    nw = struct.unpack("<i", b[at:at + 4])[0]
    old = np.frombuffer(b, "<u2", nw, at + 4).reshape(hd[6], hd[2])
    rng = np.random.default_rng(seed)
    new = np.zeros((hd[6], ne), "<u2")
    k = min(ne, hd[2])
    new[:, :k] = old[:, :k]
    if ne > k:
        new[:, k:] = rng.integers(0, hd[4], size=(hd[6], ne - k))
    hd[2] = ne
    out = b[:hd_at] + struct.pack("<10i", *hd) + b[hd_at + 40:at] + struct.pack("<i", new.size) + new.tobytes()
    with open(dst, "wb") as fh:
        fh.write(out)


def _write_tmat(dst, n_tmat, ne, seed):
    from tests.test_cabi_host import _write_s3
    rng = np.random.default_rng(seed)
    tm = np.zeros((n_tmat, ne, ne + 1), "<f4")
    for i in range(n_tmat):
        kind = i % 4  # 0: +2 skips everywhere, 1: some, 2: a state without a self-loop, 3: plain
        for j in range(ne):
            tm[i, j, j] = rng.uniform(0.3, 0.8)
            tm[i, j, j + 1] = rng.uniform(0.2, 0.6)
            if j + 2 <= ne and (kind == 0 or (kind == 1 and rng.random() < 0.5)):
                tm[i, j, j + 2] = rng.uniform(0.05, 0.5)
            if ne != 5 and j + 3 <= ne and kind == 0:
                tm[i, j, j + 3] = rng.uniform(0.02, 0.2)
        if kind == 2 and ne not in (3, 5) and ne > 1:
            tm[i, ne - 1, ne - 1] = 0.0  # the generic evaluator asks whether the arc exists
    _write_s3(dst, struct.pack("<4i", n_tmat, ne, ne + 1, tm.size) + tm.tobytes())


def _models(tmp_path, oracle_mod, ne):
    """GPU model and oracle of n_emit = ne.  4 and 5 states: an mdef of that shape.  1 and 2:
    cutting en-us's senone sequences would leave senones that no phone owns (the loader refuses
    such an mdef: the reference would index its codebook table with -1), so the mdef stays
    en-us's and only the transition matrices have the new shape -- ssw_align_batch takes its
    senone ids and matrix ids from the caller, and the oracle gets the matrices as an override."""
    src = ssw.model_dir("en-us")
    mdef, tmat = str(tmp_path / f"mdef{ne}"), str(tmp_path / f"tmat{ne}")
    if ne > 3:
        _write_mdef(mdef, ne, 100 + ne)
    else:
        mdef = os.path.join(src, "mdef")
    _write_tmat(tmat, 42, ne, 200 + ne)
    kw = dict(mdef=mdef, means=os.path.join(src, "means"), sendump=os.path.join(src, "sendump"))
    g = ssw.Model(variances=os.path.join(src, "variances"), tmat=tmat, **kw)
    assert g.tmat_n_emit == ne
    tp = g.table("tp").reshape(42, ne, ne + 1)
    if ne > 3:
        o = oracle_mod.Model(vars=os.path.join(src, "variances"), tmat=tmat, **kw)
        assert g.n_emit_state == ne and o.sseq.shape[1] == ne
        assert np.array_equal(tp, o.tp)
        assert np.array_equal(g.table("sseq"), o.sseq.reshape(-1))
    else:
        o = oracle_mod.Model(vars=os.path.join(src, "variances"),
                             tmat=os.path.join(src, "transition_matrices"), **kw)
        # the loaders agree on 3-state files (tests/test_cabi_host.py); here the same arithmetic
        # runs over rows of another length: check a few entries by hand (tmat.c:206)
        assert (tp[:, np.arange(ne), np.arange(ne)] < 255).any()
    return g, o, tp


def _task(o, ne, n_phones, seed):
    senid, tmat, ssid = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone,
                                             n_phones, seed)
    return np.ascontiguousarray(senid[:, :ne]), tmat, ssid


def _senscr(n_frames, n_sen, seed):
    scr = np.floor(lcg_uniform(seed, n_frames * n_sen).reshape(n_frames, n_sen) * 600).astype(np.int16)
    scr[np.arange(n_frames), np.floor(lcg_uniform(seed + 1, n_frames) * n_sen).astype(int)] = 0
    return scr


@pytest.mark.parametrize("ne", [5, 4, 2, 1])
def test_other_topologies_match_oracle(oracle_mod, tmp_path, ne):
    g, o, tp = _models(tmp_path, oracle_mod, ne)
    # one utterance at a time, across the 64-phone word boundaries of the kernel
    for n_phones, n_frames in ((1, 7), (6, 60), (64, 500), (65, 520), (150, 1200)):
        senid, tmat, _ = _task(o, ne, n_phones, 31 * ne + n_phones)
        assert senid.shape == (n_phones, ne)
        scr = _senscr(n_frames, o.n_sen, 5 * ne + n_phones)
        rv, rst, _ = o.state_align(scr, senid, tmat, tp=tp)
        d = g.to_device(scr)
        try:
            st, status = g.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat)
        finally:
            g.device_free(d)
        assert (status[0] == 0) == (rv == 0), (ne, n_phones, status[0], rv)
        assert rv == 0, (ne, n_phones)
        assert st.shape == (n_phones * ne, 3)
        assert np.array_equal(st, rst), (ne, n_phones)


@pytest.mark.parametrize("ne", [5, 4])
def test_other_topologies_ragged_batch_windows_and_failures(oracle_mod, tmp_path, ne):
    """Several utterances in one launch: word-like windows on one, one too short to reach its
    last phone (status -1), one frame, one whose windows leave a frame without a token."""
    g, o, tp = _models(tmp_path, oracle_mod, ne)
    shapes = [(3, 40), (70, 700), (1, 1), (30, 20), (130, 1100), (12, 200)]
    frame_off, phone_off = [0], [0]
    senids, tmats, sfs, efs, scrs, refs = [], [], [], [], [], []
    for i, (n_ph, n_fr) in enumerate(shapes):
        senid, tmat, _ = _task(o, ne, n_ph, 1000 + 7 * ne + i)
        scr = _senscr(n_fr, o.n_sen, 2000 + i)
        sf = np.zeros(n_ph, np.int32)
        ef = np.full(n_ph, 2**31 - 1, np.int32)
        if i == 1:
            sf[35:] = 350
            ef[:35] = 350
        if i == 5:      # nothing may be entered between frames 60 and 120: the search dies there
            sf[4:] = 120
            ef[:4] = 60
        refs.append(o.state_align(scr, senid, tmat, sf=sf, ef=ef, tp=tp))
        senids.append(senid); tmats.append(tmat); sfs.append(sf); efs.append(ef); scrs.append(scr)
        frame_off.append(frame_off[-1] + n_fr)
        phone_off.append(phone_off[-1] + n_ph)
    assert any(r[0] != 0 for r in refs) and any(r[0] == 0 for r in refs)
    d = g.to_device(np.concatenate(scrs))
    try:
        st, status = g.align_batch(d, frame_off, phone_off, np.concatenate(senids),
                                   np.concatenate(tmats), sf=np.concatenate(sfs),
                                   ef=np.concatenate(efs))
    finally:
        g.device_free(d)
    for u, (rv, rst, _) in enumerate(refs):
        assert (status[u] == 0) == (rv == 0), (u, status[u], rv)
        if rv == 0:
            assert np.array_equal(st[phone_off[u] * ne:phone_off[u + 1] * ne], rst), u


def test_renormalisation_with_five_states(oracle_mod, tmp_path):
    """Scores that fall fast enough for state_align_search.c:196's renormalisation to fire
    (hmm_normalize over five states and the exit score)."""
    g, o, tp = _models(tmp_path, oracle_mod, 5)
    n_phones, n_frames = 4, 17500
    senid, tmat, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone,
                                          n_phones, 77)
    scr = np.full((n_frames, o.n_sen), 32000, np.int16)
    scr[:, ::7] = 31000
    rv, rst, _, trace = o.state_align(scr, senid, tmat, want_trace=True)
    assert rv == 0
    assert trace.min() - 0x300000 < -536870912, "test must actually reach the renormalisation"
    d = g.to_device(scr)
    try:
        st, status = g.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat)
    finally:
        g.device_free(d)
    assert status[0] == 0 and np.array_equal(st, rst)


def test_state_align_search_object_with_five_states(oracle_mod, tmp_path, means_en):
    """init / start / step / finish (src/decoder.c:777-795) on a 5-state model: scoring, the
    alignment and alignment_propagate's phone entries."""
    g, o, tp = _models(tmp_path, oracle_mod, 5)
    n_ph, n_fr = 9, 140
    senid, tmat, ssid = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone,
                                             n_ph, 31)
    feats = synth_features(means_en, n_fr, 31)
    mg = ssw.PtmMgau(g)
    s = ssw.StateAlignSearch(g, mg, ssid, tmat)
    s.start()
    for t in range(n_fr):
        s.step(feats[t], t)
    s.finish()
    scr = o.ptm_score_utt(feats)
    rv, rst, rph = o.state_align(scr, senid, tmat)
    assert rv == 0
    assert np.array_equal(s.states(), rst)
    assert np.array_equal(s.phones(), rph)
    s.free()
    mg.free()


def test_what_stays_three_state_says_so(oracle_mod, tmp_path):
    """The first pass, compact rows and the active-set call are built for 3-state HMMs: a model
    of another shape is refused there with a message, not misread."""
    g, o, tp = _models(tmp_path, oracle_mod, 5)
    senid, tmat, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone, 3, 3)
    with pytest.raises(ssw.SswError, match="3-state"):
        g.compact_plan([0, 10], [0, 5], senid)      # (15 ids: read as 5 phones of 3 states)
