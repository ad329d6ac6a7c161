"""GPU parity: PTM senone scoring through the C ABI vs the CPU oracle (bit-exact int16)."""
import numpy as np
import pytest

from soundswallower_amd.synth import synth_features

pytestmark = pytest.mark.gpu


def _oracle_batch(orc, feats, utt_off):
    outs, cws, scs = [], [], []
    for u in range(len(utt_off) - 1):
        o, cw, sc = orc.ptm_score_utt(feats[utt_off[u]:utt_off[u + 1]], want_topn=True)
        outs.append(o)
        cws.append(cw)
        scs.append(sc)
    return np.concatenate(outs), np.concatenate(cws), np.concatenate(scs)


def test_tables_match_oracle(gpu_en, orc_en):
    for name, ref in (("mean", orc_en.mean), ("var", orc_en.var), ("det", orc_en.det.reshape(-1)),
                      ("ptm_mixw", orc_en.ptm_mixw.reshape(-1)), ("tp", orc_en.tp.reshape(-1)),
                      ("sseq", orc_en.sseq.reshape(-1)), ("sen2cb", orc_en.sen2cimap),
                      ("logadd8", orc_en.logadd_table_8b.astype(np.uint8))):
        got = gpu_en.table(name)
        assert got.shape == ref.shape, name
        assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(ref).view(np.uint8)), name


def test_ptm_small_batch_bit_exact(gpu_en, orc_en, means_en):
    lens = [40, 1, 17, 64]
    utt_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = np.concatenate([synth_features(means_en, n, 12345 + i) for i, n in enumerate(lens)])
    got = gpu_en.score_batch(feats, utt_off)
    ref, rcw, rsc = _oracle_batch(orc_en, feats, utt_off)
    gcw, gsc = gpu_en.last_topn(len(feats))
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert got.dtype == np.int16 and got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_ptm_config2_layouts_bit_exact(gpu_en, orc_en, means_en):
    """BASELINE config 2: 4096 frames as 16 x 256 and as 1 x 4096, full int16 parity, and the
    top-N state (codeword order) of every frame."""
    feats = np.concatenate([synth_features(means_en, 256, 12345 + i) for i in range(16)])
    for utt_off in (np.arange(17, dtype=np.int32) * 256, np.array([0, 4096], np.int32)):
        got = gpu_en.score_batch(feats, utt_off)
        flagged, pairs = gpu_en.last_stats()
        gcw, _ = gpu_en.last_topn(len(feats))
        ref, rcw, _ = _oracle_batch(orc_en, feats, utt_off)
        assert pairs == 4096 * 126
        assert 0 < flagged < pairs // 20, "exact pass should be rare but not empty here"
        assert np.array_equal(gcw.astype(np.int32), rcw)
        assert np.array_equal(got, ref)


def test_ptm_correlated_frames_bit_exact(gpu_en, orc_en, means_en):
    """Slowly drifting features (speech-like): carried top-N codewords stay good, the regime in
    which the reference's history matters most."""
    base = synth_features(means_en, 8, 777)
    t = np.linspace(0, 1, 300, dtype=np.float32)[:, None]
    feats = (base[0] * (1 - t) + base[1] * t).astype(np.float32)
    feats = np.concatenate([feats, np.repeat(base[2:3], 40, axis=0)])  # identical frames too
    got = gpu_en.score_batch(feats)
    ref = orc_en.ptm_score_utt(feats)
    assert np.array_equal(got, ref)


def test_ptm_scan_bound_stress(gpu_en, orc_en, means_en):
    """Inputs chosen against the speculative scan's quadratic form and its error bound: features
    far from every mean (large cancellation in a*x + b*x^2), tiny features, frames sitting on
    the means of the ill-conditioned densities the scan leaves to the exact form (floored
    variances), midpoints between two densities of a codebook (near-ties between candidates),
    and exact repeats.  All of it must still be bit-exact, top-N order included."""
    rng = np.random.default_rng(2024)
    base = synth_features(means_en, 64, 4321)
    var = orc_en.var.reshape(means_en.shape)
    parts = [base * 8.0, base * 40.0, base * 0.01, -base, np.zeros((4, 39), np.float32)]
    # frames on (and a hair off) the means of the densities with the largest precision terms
    flat = np.argsort(var.max(axis=3).reshape(-1))[-48:]
    on_mean = np.empty((len(flat), 39), np.float32)
    for i, k in enumerate(flat):
        cb, f, d = np.unravel_index(k, var.shape[:3])
        row = base[i % len(base)].copy()
        row[f * 13:(f + 1) * 13] = means_en[cb, f, d]
        on_mean[i] = row
    parts += [on_mean, on_mean + np.float32(1e-3), on_mean * np.float32(1.0001)]
    # midpoints of density pairs inside one codebook, all three streams at once
    mid = np.empty((64, 39), np.float32)
    for i in range(64):
        cb = int(rng.integers(0, means_en.shape[0]))
        for f in range(3):
            d1, d2 = rng.choice(means_en.shape[2], 2, replace=False)
            mid[i, f * 13:(f + 1) * 13] = (means_en[cb, f, d1] + means_en[cb, f, d2]) * 0.5
    parts += [mid, np.repeat(mid[:3], 5, axis=0)]
    feats = np.ascontiguousarray(np.concatenate(parts), np.float32)
    assert np.isfinite(feats).all()
    got = gpu_en.score_batch(feats)
    flagged, pairs = gpu_en.last_stats()
    gcw, _ = gpu_en.last_topn(len(feats))
    ref, rcw, _ = orc_en.ptm_score_utt(feats, want_topn=True)
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert np.array_equal(got, ref)
    assert flagged < pairs // 4, "the proof should still cover most pairs on these inputs"


def test_ptm_large_batch_is_size_independent(gpu_en, orc_en, means_en):
    """65,536 frames in one call (256 utterances x 256): every utterance's scores equal what the
    same utterance gets in a 1-utterance call (utterances are independent: reset history), a
    sample is checked against the oracle, and the exact pass stays rare."""
    n_utt, n_fr = 256, 256
    feats = np.concatenate([synth_features(means_en, n_fr, 9000 + u) for u in range(n_utt)])
    off = (np.arange(n_utt + 1) * n_fr).astype(np.int32)
    big = gpu_en.score_batch(feats, off)
    flagged, pairs = gpu_en.last_stats()
    assert pairs == n_utt * n_fr * 126 and flagged < pairs // 100
    assert big.shape == (n_utt * n_fr, orc_en.n_sen)
    for u in (0, 1, 97, 255):
        one = gpu_en.score_batch(feats[off[u]:off[u + 1]])
        assert np.array_equal(big[off[u]:off[u + 1]], one), u
    ref = orc_en.ptm_score_utt(feats[off[97]:off[98]])
    assert np.array_equal(big[off[97]:off[98]], ref)
    # checksum of checksums: rows are already best-score normalised, so every row holds a zero
    assert (big.max(axis=1) <= 0).all() or (big.min(axis=1) == 0).all()


@pytest.mark.parametrize("ds", [2, 3])
def test_ptm_frame_downsampling_bit_exact(oracle_mod, means_en, ds):
    """ds > 1 (src/ptm_mgau.c:241: codebooks are re-scanned only every ds-th frame, the frames in
    between re-score the carried codewords) makes every frame depend on its predecessor: the batch
    call takes the exact chain kernel.  Ragged batch, top-N order included."""
    import soundswallower_amd as ssw
    mdir = ssw.model_dir("en-us")
    g = ssw.Model(mdir, config={"ds": ds})
    o = oracle_mod.Model(mdir, config={"ds": ds})
    lens = [37, 1, 64, 130]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = np.concatenate([synth_features(means_en, n, 555 + i) for i, n in enumerate(lens)])
    got = g.score_batch(feats, off)
    gcw, _ = g.last_topn(len(feats))
    ref, rcw, _ = _oracle_batch(o, feats, off)
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert np.array_equal(got, ref)


def test_ptm_forced_exact_path_bit_exact(orc_en, means_en, monkeypatch):
    """SSW_PTM_EXACT=1 scores whole batches with the exact chain kernel (no speculation): the same
    scores and top-N state as the speculative path and the oracle."""
    import soundswallower_amd as ssw
    monkeypatch.setenv("SSW_PTM_EXACT", "1")
    g = ssw.Model(ssw.model_dir("en-us"))
    monkeypatch.delenv("SSW_PTM_EXACT")
    lens = [200, 56]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = np.concatenate([synth_features(means_en, n, 31 + i) for i, n in enumerate(lens)])
    got = g.score_batch(feats, off)
    flagged, pairs = g.last_stats()
    assert flagged == pairs == len(feats) * 126      # every pair took the exact route
    ref, rcw, _ = _oracle_batch(orc_en, feats, off)
    gcw, _ = g.last_topn(len(feats))
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert np.array_equal(got, ref)


def test_ptm_clustered_4bit_sendump_bit_exact(oracle_mod, orc_en, means_en, tmp_path):
    """A 4-bit clustered sendump: the product expands it once at load, the oracle decodes it per
    lookup the way the reference does (nibble chosen by the packed byte's own low bit,
    src/ptm_mgau.c:375-378); the scores must agree."""
    import os
    import soundswallower_amd as ssw
    from tests.test_cabi_host import synth_clustered_sendump
    src = ssw.model_dir("en-us")
    sd = str(tmp_path / "sendump4")
    synth_clustered_sendump(orc_en, sd, seed=9)
    kw = dict(mdef=os.path.join(src, "mdef"), means=os.path.join(src, "means"), sendump=sd,
              tmat=os.path.join(src, "transition_matrices"))
    g = ssw.Model(variances=os.path.join(src, "variances"), **kw)
    o = oracle_mod.Model(vars=os.path.join(src, "variances"), **kw)
    feats = synth_features(means_en, 300, 2718)
    assert np.array_equal(g.score_batch(feats), o.ptm_score_utt(feats))


def test_ptm_edge_shapes(gpu_en, orc_en, means_en):
    """Empty batch, empty utterances inside a batch, one-frame utterances, many tiny utterances:
    the shapes the reference's tests call 'empty and ragged inputs'."""
    assert gpu_en.score_batch(np.zeros((0, 39), np.float32)).shape == (0, orc_en.n_sen)
    feats = synth_features(means_en, 40, 99)
    # utterances of 0, 1, 0, 38, 1, 0 frames
    off = np.array([0, 0, 1, 1, 39, 40, 40], np.int32)
    got = gpu_en.score_batch(feats, off)
    ref = np.concatenate([orc_en.ptm_score_utt(feats[off[u]:off[u + 1]])
                          for u in range(len(off) - 1) if off[u + 1] > off[u]])
    assert np.array_equal(got, ref)
    # 600 one-frame utterances: every frame starts from the reset history
    f1 = synth_features(means_en, 600, 100)
    got1 = gpu_en.score_batch(f1, np.arange(601, dtype=np.int32))
    ref1 = np.concatenate([orc_en.ptm_score_utt(f1[t:t + 1]) for t in range(600)])
    assert np.array_equal(got1, ref1)
    with pytest.raises(Exception):
        gpu_en.score_batch(feats, np.array([0, 41], np.int32))      # offsets past the batch
    with pytest.raises(Exception):
        gpu_en.score_batch(feats, np.array([0, 30, 20, 40], np.int32))  # not ascending


def _unprovable_pairs(orc, means, x):
    """[n][n_cb][n_feat] bool: pairs whose top-N list may depend on the carried history -- the
    four best truncated densities are not four distinct ints above every other density's
    (DESIGN.md section 4).  Reference arithmetic in numpy float32 (one rounding per operation)."""
    n_cb, n_feat, n_den, vl = means.shape
    mean = orc.mean.reshape(means.shape)
    var = orc.var.reshape(means.shape)
    det = orc.det.reshape(n_cb, n_feat, n_den)
    x = np.ascontiguousarray(x, np.float32).reshape(len(x), 1, n_feat, 1, vl)
    d = np.broadcast_to(det, (x.shape[0],) + det.shape).astype(np.float32).copy()
    for j in range(vl):
        diff = x[..., j] - mean[None, ..., j]
        sq = diff * diff
        d = d - sq * var[None, ..., j]
    iv = np.trunc(np.maximum(d, np.float32(-2147483648.0))).astype(np.int64)
    top = -np.sort(-iv, axis=-1)[..., :5]
    distinct = (top[..., 0] > top[..., 1]) & (top[..., 1] > top[..., 2]) \
        & (top[..., 2] > top[..., 3]) & (top[..., 3] > top[..., 4])
    return ~distinct


@pytest.mark.parametrize("total", [2600, 900])
def test_ptm_runs_of_history_dependent_frames_across_tiles(gpu_en, orc_en, means_en, total):
    """The exact in-wave pass of the frames kernel under the worst inputs for it: long runs of
    IDENTICAL frames (digital silence does this to real audio) whose top-N lists tie, so that
    every frame of a run depends on its predecessor.  Runs longer than a tile (128 frames at 2
    frames per lane, 64 at 1) make waves re-derive the order their tile starts from by walking
    back over frames they do not own; runs across the boundary between a lane's first and second
    frame, runs that start mid-tile, and an utterance boundary inside a run (reset history) are
    all here.  `total` selects the kernel variant: 2 frames per lane from ~2100 frames up."""
    rng = np.random.default_rng(42 + total)
    cand = synth_features(means_en, 600, 31337)
    cand = (np.round(cand * 8.0) / 8.0).astype(np.float32)     # coarse grid: more ties
    bad = _unprovable_pairs(orc_en, means_en, cand)
    idx = np.argsort(-bad.reshape(len(cand), -1).sum(axis=1))[:6]
    assert bad[idx[0]].any(), "no candidate frame with a history-dependent pair"
    ties = cand[idx]
    filler = synth_features(means_en, 64, 999)
    parts, lens = [], []
    pos = 0
    runs = [3, 331, 17, 140, 1, 129, 65, 64, 200]
    k = 0
    while pos < total:
        n = runs[k % len(runs)]
        if k % 2 == 0:
            blk = filler[rng.integers(0, len(filler), n)]
        else:
            blk = np.repeat(ties[(k // 2) % len(ties)][None], n, axis=0)
        parts.append(blk)
        pos += n
        k += 1
    feats = np.ascontiguousarray(np.concatenate(parts)[:total], np.float32)
    # utterance boundaries: one inside the first long run, one at a tile edge, one ragged
    off = np.array([0, 150, 384, 385, total], np.int32)
    got = gpu_en.score_batch(feats, off)
    flagged, pairs = gpu_en.last_stats()
    gcw, _ = gpu_en.last_topn(len(feats))
    ref, rcw, _ = _oracle_batch(orc_en, feats, off)
    assert pairs == total * 126
    assert flagged >= 300, "the runs should put many pairs through the exact pass"
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert np.array_equal(got, ref)
    if total >= 2600:
        # round 3: large batches are scored in pieces (scan + senone per ~16 K frames); pieces of
        # 512 and 768 frames here, so that their edges fall inside the runs: a piece's first
        # frame re-derives the order it starts from by walking back into the piece before
        import os
        for piece in ("512", "768"):
            os.environ["SSW_SCORE_PIECE"] = piece
            try:
                got2 = gpu_en.score_batch(feats, off)
                f2, p2 = gpu_en.last_stats()
                gcw2, _ = gpu_en.last_topn(len(feats))
            finally:
                del os.environ["SSW_SCORE_PIECE"]
            assert p2 == pairs and f2 == flagged, piece
            assert np.array_equal(gcw2.astype(np.int32), rcw), piece
            assert np.array_equal(got2, ref), piece


def _tie_heavy_features(orc, means, total, seed):
    """runs of identical frames whose top-N lists tie (see the test above), so that many frames
    depend on the carried history"""
    rng = np.random.default_rng(seed)
    cand = synth_features(means, 600, 31337)
    cand = (np.round(cand * 8.0) / 8.0).astype(np.float32)
    bad = _unprovable_pairs(orc, means, cand)
    ties = cand[np.argsort(-bad.reshape(len(cand), -1).sum(axis=1))[:6]]
    filler = synth_features(means, 64, 999)
    parts, pos, k = [], 0, 0
    runs = [3, 331, 17, 140, 1, 129, 65, 64, 200]
    while pos < total:
        n = runs[k % len(runs)]
        parts.append(filler[rng.integers(0, len(filler), n)] if k % 2 == 0
                     else np.repeat(ties[(k // 2) % len(ties)][None], n, axis=0))
        pos += n
        k += 1
    return np.ascontiguousarray(np.concatenate(parts)[:total], np.float32)


def test_history_carried_between_calls_and_utterances(gpu_en, orc_en, means_en):
    """ssw_score_batch_ex: the reference never resets its top-N history after start-up
    (src/acmod.c:367, src/ptm_mgau.c:425-448).  (i) a chain scored in two calls, the second
    starting from the first's carry_out, equals the chain scored in one piece, cut inside a run
    of history-dependent frames; (ii) SSW_SCORE_CARRY_UTTS: utterance boundaries do not reset;
    (iii) the plain call still resets.  Large calls (matrix-core scan) and small ones."""
    feats = _tie_heavy_features(orc_en, means_en, 4700, 7)
    ref_chain = orc_en.ptm_score_utt(feats)
    for cut in (2400, 170):                 # the cut lies inside a run of identical frames
        a, carry = gpu_en.score_batch_carry(feats[:cut])
        b, carry2 = gpu_en.score_batch_carry(feats[cut:], carry_in=carry)
        assert np.array_equal(np.concatenate([a, b]), ref_chain), cut
        assert carry2.shape == (126,) and (carry2 != 0x03020100).any()
    # without the carry the second piece starts from the reset history: the oracle's restart
    b_reset, _ = gpu_en.score_batch_carry(feats[2400:])
    assert np.array_equal(b_reset, orc_en.ptm_score_utt(feats[2400:]))
    # utterance boundaries: reset by default, carried with the flag
    off = np.array([0, 150, 2400, 2400, 4700], np.int32)
    carried, _ = gpu_en.score_batch_carry(feats, off, carry_utts=True)
    assert np.array_equal(carried, ref_chain)
    plain = gpu_en.score_batch(feats, off)
    ref_reset = np.concatenate([orc_en.ptm_score_utt(feats[off[u]:off[u + 1]])
                                for u in range(len(off) - 1) if off[u + 1] > off[u]])
    assert np.array_equal(plain, ref_reset)
    # round 3: the same calls with the batch scored in pieces (large batches are; here pieces of
    # 512 frames, whose edges fall inside the runs): carried chains, carry_in and resets alike
    import os
    os.environ["SSW_SCORE_PIECE"] = "512"
    try:
        carried2, _ = gpu_en.score_batch_carry(feats, off, carry_utts=True)
        a2, carry = gpu_en.score_batch_carry(feats[:2400])
        b2, _ = gpu_en.score_batch_carry(feats[2400:], carry_in=carry)
        plain2 = gpu_en.score_batch(feats, off)
    finally:
        del os.environ["SSW_SCORE_PIECE"]
    assert np.array_equal(carried2, ref_chain)
    assert np.array_equal(np.concatenate([a2, b2]), ref_chain)
    assert np.array_equal(plain2, ref_reset)


def _ring_features(orc, means, total):
    """Utterances whose boundaries show the reference's two-slot history ring: triplets (y, z, x)
    of frames, found with the oracle, such that x scored after y differs from x scored after z.
    `... y z | x ...` with y at an odd frame number and z the last frame of its utterance (odd
    length): the next utterance's frame 0 copies slot 1 = y's order, not z's
    (src/ptm_mgau.c:425-437).  Also `... y | z | x ...` (a one-frame utterance writes slot 0 only)
    and an utterance without frames.  Filled up to `total` frames with the tie-heavy runs."""
    cand = (np.round(synth_features(means, 600, 31337) * 8.0) / 8.0).astype(np.float32)
    bad = _unprovable_pairs(orc, means, cand).reshape(len(cand), -1).sum(axis=1)
    xs = np.argsort(-bad)[:8]
    trip = []
    for xi in xs:
        rows = {}
        for yi in range(0, 60):
            r = orc.ptm_score_utt(np.stack([cand[yi], cand[xi]]))[1]
            rows.setdefault(r.tobytes(), []).append(yi)
        groups = sorted(rows.values(), key=len)
        if len(groups) >= 2:
            trip.append((groups[0][0], groups[-1][0], xi))      # (y, z, x): x differs after them
    assert len(trip) >= 3, "no history-sensitive triplets among the candidates"
    filler = synth_features(means, 64, 999)
    parts, lens = [], []

    def utt(*rows):
        a = np.stack(rows) if rows else np.zeros((0, 39), np.float32)
        parts.append(a.astype(np.float32))
        lens.append(len(a))
    k = 0
    for (y, z, x) in trip:
        f = [filler[(k + i) % 64] for i in range(5)]
        utt(f[0], cand[y], cand[z])                 # odd: slot 1 = y
        utt(cand[x], f[1], f[2], cand[y])           # even, ends on y (odd number)
        utt(cand[z])                                # one frame: slot 1 still y
        utt()                                       # no frame
        utt(cand[z])                                # another one
        utt(cand[x], f[3], cand[y], cand[z], f[4])  # odd (5): slot 1 = z at number 3
        utt(cand[x], cand[y])
        k += 5
    rest = total - sum(lens)
    tie = _tie_heavy_features(orc, means, rest, 11)
    cut = [c for c in (0, 3, 334, 335, 351, 492) if c < rest] + [rest]
    for a, b in zip(cut[:-1], cut[1:]):
        parts.append(tie[a:b])
        lens.append(b - a)
    feats = np.ascontiguousarray(np.concatenate(parts), np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return feats, off


def test_host_call_cuts_long_utterances_and_hands_the_history_on(gpu_en, orc_en, means_en,
                                                                 monkeypatch):
    """ssw_score_batch_host used to size its pinned staging and device rows by the
    longest utterance.  An utterance beyond SSW_HOST_PIPE_CAP frames (16,384; 300 here) is now
    scored in pieces that carry the top-N history (carry_out -> carry_in), between short
    utterances that are grouped as before: the rows must be those of the uncut utterances --
    tie-heavy frames, so a piece that started from the reset history would differ -- and the
    whole-call top-N view must cover every piece."""
    feats = _tie_heavy_features(orc_en, means_en, 2500, 11)
    # 1000 and 1300 frames > cap; the first one's third piece starts at frame 700 of the chain,
    # inside a run of tied frames where the order carried in decides (checked with the oracle)
    off = np.array([0, 100, 1100, 1100, 1160, 1161, 2461, 2500], np.int32)
    ref = np.concatenate([orc_en.ptm_score_utt(feats[off[u]:off[u + 1]])
                          for u in range(len(off) - 1) if off[u + 1] > off[u]])
    whole = gpu_en.score_batch(feats, off)
    assert np.array_equal(whole, ref)
    cw_whole, _ = gpu_en.last_topn(len(feats))
    monkeypatch.setenv("SSW_HOST_PIPE_CAP", "300")
    cut = gpu_en.score_batch(feats, off)
    assert np.array_equal(cut, ref)
    cw_cut, _ = gpu_en.last_topn(len(feats))
    assert np.array_equal(cw_cut, cw_whole)
    # the carried history matters on these inputs: the same pieces from the reset history differ
    pieces_reset = np.concatenate([gpu_en.score_batch(feats[a:min(a + 300, 1100)])
                                   for a in range(100, 1100, 300)])
    assert not np.array_equal(pieces_reset, ref[100:1100])
    monkeypatch.setenv("SSW_HOST_PIPE_CAP", "301")        # odd values are rounded down to even
    assert np.array_equal(gpu_en.score_batch(feats, off), ref)


def test_chain_over_utterances_follows_the_two_slot_ring(gpu_en, orc_en, means_en, monkeypatch):
    """the reference's history is a ring of two slots indexed by frame % 2 and frame
    numbers restart with every utterance (src/ptm_mgau.c:425-437, src/acmod.c:367), so frame 0
    of an utterance starts from the last ODD-numbered frame before it: frame T - 2 of a
    predecessor of odd length T, further back across one-frame utterances.  The boundaries are
    built so that the choice shows (_ring_features); the oracle scores the utterances one after
    the other with its own ring.  All three top-N kernels, and carry_in / carry_out with
    SSW_SCORE_CARRY_OUT_REWIND across two calls."""
    feats, off = _ring_features(orc_en, means_en, 4700)
    want = orc_en.ptm_score_chain(feats, off)
    one_utt = orc_en.ptm_score_utt(feats)
    differ = np.nonzero((want != one_utt).any(axis=1))[0]
    assert len(differ) >= 6, differ               # the ring matters on this input
    got, _ = gpu_en.score_batch_carry(feats, off, carry_utts=True)            # matrix-core scan
    assert np.array_equal(got, want)
    monkeypatch.setenv("SSW_SCAN", "fma")
    got, _ = gpu_en.score_batch_carry(feats, off, carry_utts=True)            # vector-unit scan
    monkeypatch.delenv("SSW_SCAN")
    assert np.array_equal(got, want)
    monkeypatch.setenv("SSW_SCORE_PIECE", "512")
    got, _ = gpu_en.score_batch_carry(feats, off, carry_utts=True)            # pieces
    monkeypatch.delenv("SSW_SCORE_PIECE")
    assert np.array_equal(got, want)
    # two calls (small ones: vector-unit scan, one frame per lane): the first hands slot 1 over
    # with SSW_SCORE_CARRY_OUT_REWIND, whatever the lengths of its last utterances
    for cut_u in range(1, 16):
        cut = int(off[cut_u])
        a, carry = gpu_en.score_batch_carry(feats[:cut], off[:cut_u + 1], carry_utts=True,
                                            rewind=True)
        b, _ = gpu_en.score_batch_carry(feats[cut:], off[cut_u:] - cut, carry_in=carry,
                                        carry_utts=True)
        assert np.array_equal(np.concatenate([a, b]), want), cut_u
    # a first call of one-frame utterances only hands its carry_in through
    mark = np.full(126, 0x05040302, np.uint32)
    _, carry = gpu_en.score_batch_carry(feats[:2], [0, 1, 2], carry_in=mark, carry_utts=True,
                                        rewind=True)
    assert np.array_equal(carry, mark)
    _, carry = gpu_en.score_batch_carry(feats[:1], rewind=True)
    assert (carry == 0x03020100).all()


def test_chain_over_utterances_exact_kernel(orc_en, means_en, monkeypatch):
    """the same chain through the sequential kernel (SSW_PTM_EXACT=1, what ds != 1 uses)"""
    import soundswallower_amd as ssw
    feats, off = _ring_features(orc_en, means_en, 900)
    want = orc_en.ptm_score_chain(feats, off)
    monkeypatch.setenv("SSW_PTM_EXACT", "1")
    m = ssw.Model(ssw.model_dir("en-us"))
    monkeypatch.delenv("SSW_PTM_EXACT")
    got, _ = m.score_batch_carry(feats, off, carry_utts=True, rewind=True)
    assert np.array_equal(got, want)
    cut = int(off[8])
    a, c1 = m.score_batch_carry(feats[:cut], off[:9], carry_utts=True, rewind=True)
    b, _ = m.score_batch_carry(feats[cut:], off[8:] - cut, carry_in=c1, carry_utts=True)
    assert np.array_equal(np.concatenate([a, b]), want)


def test_features_beyond_the_binary16_range_take_the_exact_pass(gpu_en, orc_en, means_en):
    """Round 4: the matrix-core scan cuts x and x^2 into binary16 parts, so a feature beyond
    +-255 (its square does not fit) makes every key of its frame -inf or NaN; the kernel does not
    trust such a frame and redoes it exactly -- with a floor for the exact scan only when the
    four candidates are four different codewords.  Frames at, just below and far beyond the
    limit, in single dimensions and all over, mixed into a batch large enough for the
    matrix-core scan; scores and top-N order against the oracle, and only the frames that need it
    (plus the usual handful) take the exact route."""
    feats = np.concatenate([synth_features(means_en, 256, 4242 + i) for i in range(10)])
    rng = np.random.default_rng(7)
    hot = rng.choice(len(feats), 300, replace=False)
    vals = [255.0, -255.0, 254.99, 255.01, 256.0, -300.0, 1000.0, 65504.0, 1.0e6, -3.0e9, 1.0e19]
    beyond = np.zeros(len(feats), bool)
    for i, t in enumerate(hot):
        v = np.float32(vals[i % len(vals)])
        if i % 3 == 0:                      # one dimension of one stream
            feats[t, int(rng.integers(0, 39))] = v
        elif i % 3 == 1:                    # the same dimension of every stream
            feats[t, int(rng.integers(0, 13))::13] = v
        else:                               # a whole stream scaled up
            f = int(rng.integers(0, 3))
            feats[t, 13 * f:13 * f + 13] *= np.float32(abs(v) / 8)
        beyond[t] = np.abs(feats[t]).max() > 255.0
    assert np.isfinite(feats).all() and beyond.sum() > 150
    off = (np.arange(11) * 256).astype(np.int32)
    got = gpu_en.score_batch(feats, off)
    flagged, pairs = gpu_en.last_stats()
    gcw, _ = gpu_en.last_topn(len(feats))
    ref, rcw, _ = _oracle_batch(orc_en, feats, off)
    assert np.array_equal(gcw.astype(np.int32), rcw)
    assert np.array_equal(got, ref)
    # a frame beyond the limit in one stream sends that stream's 42 pairs through the exact pass
    assert beyond.sum() * 42 <= flagged <= beyond.sum() * 126 + pairs // 200


def test_one_and_two_step_scan_waves_agree_on_piece_sized_launches(gpu_en, orc_en, means_en, monkeypatch):
    """From 12,288 frames per launch a scan wave takes two 64-frame steps instead of one
    (csrc/ssw_host_score.inc).  32,768 frames (two pieces of 16,384): one step per wave, two steps,
    the default, four workgroups per CU instead of five (SSW_MFMA_LDS_PAD) -- identical rows,
    and a sample of utterances against the oracle.  (Round 4 shipped the two-step instance for a
    few commits with its utterance-start bits read through an inline-asm scalar load whose wait
    sat in a second statement; the compiler spilled the register pair in between.  Two rows of
    this batch differed, a different set with other occupancy.)"""
    feats = np.concatenate([synth_features(means_en, 256, 9000 + u) for u in range(128)])
    off = (np.arange(129) * 256).astype(np.int32)
    monkeypatch.setenv("SSW_MFMA_STEPS", "1")
    one = gpu_en.score_batch(feats, off)
    monkeypatch.setenv("SSW_MFMA_STEPS", "2")
    assert np.array_equal(gpu_en.score_batch(feats, off), one)
    monkeypatch.setenv("SSW_MFMA_LDS_PAD", "2048")
    assert np.array_equal(gpu_en.score_batch(feats, off), one)
    monkeypatch.delenv("SSW_MFMA_STEPS")
    assert np.array_equal(gpu_en.score_batch(feats, off), one)
    monkeypatch.delenv("SSW_MFMA_LDS_PAD")
    assert np.array_equal(gpu_en.score_batch(feats, off), one)
    for u in (0, 16, 17, 63, 64, 127):
        assert np.array_equal(one[off[u]:off[u + 1]], orc_en.ptm_score_utt(feats[off[u]:off[u + 1]])), u


def test_chain_over_utterances_with_frame_downsampling(oracle_mod, means_en):
    """ds = 2 (codebooks re-scanned every other frame, src/ptm_mgau.c:241) takes the sequential
    kernel; with SSW_SCORE_CARRY_UTTS one wave walks all utterances of a codebook x stream: the
    frame numbers -- and with them the ds phase and the history ring -- restart at every
    utterance.  Odd, one-frame and empty utterances against the oracle's chain."""
    import soundswallower_amd as ssw
    mdir = ssw.model_dir("en-us")
    g = ssw.Model(mdir, config={"ds": 2})
    o = oracle_mod.Model(mdir, config={"ds": 2})
    feats, off = _ring_features(o, means_en, 500)
    want = o.ptm_score_chain(feats, off)
    got, _ = g.score_batch_carry(feats, off, carry_utts=True)
    assert np.array_equal(got, want)
    plain = g.score_batch(feats, off)
    ref = np.concatenate([o.ptm_score_utt(feats[off[u]:off[u + 1]]) for u in range(len(off) - 1)
                          if off[u + 1] > off[u]])
    assert np.array_equal(plain, ref)
    g.close()


def test_device_mem_info(gpu_en):
    import ctypes
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert gpu_en._L.ssw_device_mem_info(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
    assert 0 < free_b.value <= total_b.value and total_b.value > (100 << 30)   # an MI355X: 288 GB
