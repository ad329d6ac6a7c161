"""ssw_align_text_batch with decoder_alignment's history semantics.

The reference scores every frame twice: once for the first pass and, after acmod_rewind, once
more for the second, the PTM top-N history carrying over the rewind (src/decoder.c:786-793,
src/ptm_mgau.c:425-448).  Where truncated densities tie, the second scoring can differ from the
first.  With ssw_first_pass_config_t.two_pass_history the one-call text path does the
same, per utterance.  Checked against a composition of calls that are each pinned to the oracle
elsewhere: first-pass scores per utterance with their carry_out (ssw_score_batch_ex), second-pass
scores from that carry (equal to the oracle's two_pass_scores, checked here too), the first pass
on the former, the constrained state alignment on the latter.  The recording is the reference's
goforward behind a few leading frames whose top-N lists tie (the two scorings can only differ from
an utterance's first frame on); a third, truncated utterance fails its first pass and must not
disturb the per-utterance rows of the others."""
import numpy as np
import pytest
import torch

import soundswallower_amd as ssw
from tests.test_gpu_first_pass import _lex
from tests.test_oracle_e2e_goforward import goforward_features, two_pass_scores

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1
TEXT = "go forward ten meters".split()


def _second_pass(gpu, lex, scr1, scr2, n):
    """first pass + populate on scr1 (forced_align_batch gives words and phone rows), then the
    state alignment of those phones, with those windows, on scr2"""
    d1 = gpu.to_device(scr1)
    try:
        a = ssw.forced_alignment(gpu, lex, d1, [0, n], [TEXT])[0]
    finally:
        gpu.device_free(d1)
    assert a is not None
    ph = lex.populate(a["words"], a["word_al"][:, 0], a["word_al"][:, 1])
    sseq = gpu.table("sseq").reshape(-1, 3)
    senid = np.ascontiguousarray(sseq[ph["ssid"]], np.uint16)
    start, dur = ph["start"], ph["duration"]
    sf = np.where(start > 0, start, 0).astype(np.int32)
    ef = np.where(dur > 0, start + dur, INT_MAX).astype(np.int32)
    init = np.stack([np.repeat(start, 3), np.repeat(dur, 3), np.zeros(3 * len(start), np.int32)], 1)
    d2 = gpu.to_device(scr2)
    try:
        st, status = gpu.align_batch(d2, [0, n], [0, len(start)], senid,
                                     np.asarray(ph["tmatid"], np.int16), sf, ef,
                                     state_init=init.astype(np.int32))
    finally:
        gpu.device_free(d2)
    assert status[0] == 0
    return a["words"], st


def test_second_pass_scored_from_the_first_pass_history(gpu_en, orc_en, oracle_mod, means_en):
    from soundswallower_amd.synth import synth_features
    plain_audio = goforward_features(oracle_mod)
    # The two scorings can only differ from the utterance's first frame on (frame t + 1 starts
    # from frame t's final order in both), and only where truncated densities tie at the edge of
    # the top-N list.  So the recording gets one leading frame x and one trailing frame y from a
    # coarse grid, chosen (with the oracle, on the CPU) so that x scored after y differs from x
    # scored from the reset history: the second scoring starts from what y left.  The first pass
    # absorbs both in its <sil>s.
    cand = (np.round(synth_features(means_en, 600, 31337) * 8.0) / 8.0).astype(np.float32)
    real = np.concatenate([cand[578:579], plain_audio, cand[169:170]]).astype(np.float32)
    coarse = np.concatenate([cand[538:539], plain_audio, cand[461:462]]).astype(np.float32)
    # the history is a ring of two slots indexed by frame % 2 (src/ptm_mgau.c:425-437),
    # and after the rewind frame 0 copies slot 1 = the last ODD-numbered frame of the first pass.
    # `real` and `coarse` have an even number of frames (that is their last frame); `odd` has one
    # more frame behind y, which the second pass must NOT start from.
    odd = np.concatenate([real, plain_audio[100:101]]).astype(np.float32)
    n = len(real)
    assert n % 2 == 0 and len(odd) % 2 == 1
    lex = _lex(gpu_en, "en-us")
    want, n_differ = [], []
    for feats in (real, coarse, odd):
        scr1, carry = gpu_en.score_batch_carry(feats, rewind=True)       # reset history
        scr2, _ = gpu_en.score_batch_carry(feats, carry_in=carry)        # from what pass 1 left
        orc_en.ptm_reset()
        assert np.array_equal(scr2, two_pass_scores(orc_en, feats))      # the reference's second scoring
        n_differ.append(int((scr1 != scr2).any(axis=1).sum()))
        want.append(_second_pass(gpu_en, lex, scr1, scr2, len(feats)))
    orc_en.ptm_reset()
    assert min(n_differ) >= 1, n_differ                     # the flag has something to change
    # the odd utterance from its LAST frame's order (round 3's rule) is not the reference's scoring
    _, last = gpu_en.score_batch_carry(odd)
    wrong, _ = gpu_en.score_batch_carry(odd, carry_in=last)
    orc_en.ptm_reset()
    assert not np.array_equal(wrong, two_pass_scores(orc_en, odd))
    orc_en.ptm_reset()
    # a one-frame utterance never writes slot 1: its second pass starts where its first did
    one = odd[:1]
    _, c1 = gpu_en.score_batch_carry(one, rewind=True)
    assert (c1 == 0x03020100).all()
    batch = np.concatenate([real, coarse, real[:40], odd, one])
    off = np.cumsum([0, n, n, 40, len(odd), 1]).astype(np.int32)
    d_feats = torch.from_numpy(batch).cuda()
    texts = [TEXT] * 5
    cfg = lex.first_pass_config(two_pass_history=1)
    aset = ssw.align_text_batch(gpu_en, lex, d_feats, off, texts, cfg=cfg)
    plain = ssw.align_text_batch(gpu_en, lex, d_feats, off, texts)
    try:
        assert aset.status(2) == 1 and plain.status(2) == 1      # 40 frames cannot hold the text
        assert aset.status(4) == 1 and plain.status(4) == 1      # nor can one
        differs_from_plain = False
        for u, w in ((0, 0), (1, 1), (3, 2)):
            got = aset.utterance(u)
            assert got is not None
            assert got["words"] == want[w][0]
            assert np.array_equal(got["state_al"], want[w][1]), u
            differs_from_plain |= not np.array_equal(plain.utterance(u)["state_al"], got["state_al"])
        # (informational: whether the tie-dependent frames lie on the best path is up to the data)
        print("two-pass history changed a state alignment:", differs_from_plain, n_differ)
    finally:
        aset.free()
        plain.free()
    # The same utterances 7 times over in one batch (5.9 K frames: the matrix-core scan), scored in
    # pieces of 1024 frames: every utterance's second pass must start from ITS OWN first pass's
    # last order (a row per utterance on the device), wherever the piece edges fall
    import os
    reps = 21
    seq = [real, coarse, odd] * (reps // 3)
    big = np.concatenate(seq)
    boff = np.cumsum([0] + [len(a) for a in seq]).astype(np.int32)
    os.environ["SSW_SCORE_PIECE"] = "1024"
    try:
        bset = ssw.align_text_batch(gpu_en, lex, torch.from_numpy(big).cuda(), boff, [TEXT] * reps,
                                    cfg=cfg)
    finally:
        del os.environ["SSW_SCORE_PIECE"]
    try:
        for u in range(reps):
            got = bset.utterance(u)
            assert got is not None and got["words"] == want[u % 3][0], u
            assert np.array_equal(got["state_al"], want[u % 3][1]), u
    finally:
        bset.free()
        lex.free()
