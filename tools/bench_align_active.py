#!/usr/bin/env python3
"""The reference's DEFAULT configuration (compallsen = no) as a batch, timed: BASELINE configs[2]'s
shape (256 utterances x 1000 frames x 150 phones) through ssw_align_batch_active(_ex) -- scoring
of the active senones only + forced alignment in one call (src/acmod.c:905-999,
src/ptm_mgau.c:297-321, 353-364, src/state_align_search.c:177-213) -- next to the compallsen = yes
pipeline (ssw_score_batch + ssw_align_batch) on the same inputs.  Two window shapes: none (every
phone is entered in frame 1: ~450 listed senones per frame from then on) and word-like windows as
alignment_populate leaves them after a first pass (the set grows along the utterance).
Prints one JSON line; bench.py's `align_default_config` object is run()."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd.synth import (read_raw_means, synth_alignment_task,  # noqa: E402
                                      synth_features)

INT_MAX = 2**31 - 1


def word_windows(rng, n_phones, n_frames):
    """groups of 1-4 phones sharing a window, windows advancing with the audio (+-8 frames)"""
    sf = np.zeros(n_phones, np.int32)
    ef = np.full(n_phones, INT_MAX, np.int32)
    i, t = 0, 0
    while i < n_phones:
        g = int(rng.integers(1, 5))
        dur = max(3 * g + 2, int(n_frames * g / n_phones))
        sf[i:i + g] = max(0, t - 8)
        ef[i:i + g] = min(n_frames, t + dur + 8)
        t += dur
        i += g
    ef[-1] = n_frames          # the last phone lives to the end of the audio
    return sf, np.maximum.accumulate(ef)


def run(model=None, scorer=None, n_utts=256, n_frames=1000, n_phones=150, reps=3, name="en-us"):
    own = model is None
    mdir = ssw.model_dir(name)
    if own:
        model = ssw.Model(mdir)
    scorer = ssw.SCORER_PTM if scorer is None else scorer
    means = read_raw_means(mdir)
    sseq = model.table("sseq").reshape(-1, 3)
    pssid, ptmat = model.table("phone_ssid"), model.table("phone_tmat")
    rng = np.random.default_rng(99)
    senid, tmat, sfw, efw = [], [], [], []
    for u in range(n_utts):
        s_, t_, _ = synth_alignment_task(sseq, pssid, ptmat, model.n_ciphone, n_phones, 777 + u)
        a, b = word_windows(rng, n_phones, n_frames)
        senid.append(s_); tmat.append(t_); sfw.append(a); efw.append(b)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    sfw, efw = np.concatenate(sfw), np.concatenate(efw)
    frame_off = (np.arange(n_utts + 1) * n_frames).astype(np.int32)
    phone_off = (np.arange(n_utts + 1) * n_phones).astype(np.int32)
    total = n_utts * n_frames
    row = model.veclen_total * 4
    d_feats = model.device_malloc(total * row)
    for u in range(n_utts):
        f = synth_features(means, n_frames, 12345 + u)
        model._L.ssw_memcpy_h2d(d_feats + u * n_frames * row, f.ctypes.data, f.nbytes)
    d_scr = model.device_malloc(total * model.n_sen * 2)
    out = {"workload": f"{n_utts} utterances x {n_frames} frames x {n_phones} phones, {name}, "
                       f"{'ms' if scorer == ssw.SCORER_MS else 'PTM'} scorer: second pass of "
                       f"forced alignment in the reference's default configuration "
                       f"(compallsen = no) as ONE ssw_align_batch_active call, against "
                       f"ssw_score_batch + ssw_align_batch (compallsen = yes) on the same inputs"}
    try:
        def best_of(fn):
            fn()
            model._L.ssw_device_synchronize()
            b = None
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                t = time.perf_counter() - t0
                b = t if b is None or t < b else b
            return b, r

        for label, sf, ef in (("no_windows", None, None), ("word_windows", sfw, efw)):
            ta, (st_a, status_a) = best_of(lambda: model.align_batch_active(
                d_feats, frame_off, phone_off, senid, tmat, sf, ef, d_senscr=d_scr, scorer=scorer))

            def yes():
                model.score_batch_device(d_feats, total, frame_off, d_scr, None, scorer=scorer)
                return model.align_batch(d_scr, frame_off, phone_off, senid, tmat, sf, ef)
            ty, (st_y, status_y) = best_of(yes)
            out[label] = {
                "active_ms": ta * 1e3, "active_utt_frames_per_s": total / ta,
                "compallsen_yes_ms": ty * 1e3, "compallsen_yes_utt_frames_per_s": total / ty,
                "aligned_active": int((status_a == 0).sum()),
                "aligned_yes": int((status_y == 0).sum()),
                # not a parity statement (the two configurations score differently by design):
                # how many utterances end with the same segmentation
                "same_segmentation": int(sum(
                    np.array_equal(st_a[phone_off[u] * 3:phone_off[u + 1] * 3, :2],
                                   st_y[phone_off[u] * 3:phone_off[u + 1] * 3, :2])
                    for u in range(n_utts)))}
    finally:
        model.device_free(d_feats)
        model.device_free(d_scr)
        if own:
            model.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--phones", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--model", default="en-us")
    ap.add_argument("--ms", action="store_true", help="the ms scorer (fr-fr with synthesised "
                    "mixture_weights, as tools/bench_ms.py)")
    a = ap.parse_args()
    if a.ms:
        import bench_ms      # fr-fr with mixture_weights synthesised from its sendump
        m, _ = bench_ms.build_model()
        r = run(m, ssw.SCORER_MS, a.utts, a.frames, a.phones, a.reps, "fr-fr")
        m.close()
    else:
        r = run(None, None, a.utts, a.frames, a.phones, a.reps, a.model)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
