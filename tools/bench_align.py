#!/usr/bin/env python3
"""BASELINE config 3 / 5 timing: N synthetic utterances x F frames, ~P phones each:
PTM scoring of every frame, then forced-alignment Viterbi of every utterance on one MI355X.
Prints one JSON line (utterance-frames/s, align RTF = wall / audio seconds at 100 frames/s)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_alignment_task, synth_features  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--phones", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    _lib.build()
    L = _lib.lib()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    means = read_raw_means(mdir)
    sseq = m.table("sseq").reshape(-1, 3)
    pssid, ptmat = m.table("phone_ssid"), m.table("phone_tmat")
    n_total = a.utts * a.frames
    feats = np.concatenate([synth_features(means, a.frames, 12345 + u) for u in range(a.utts)])
    senid, tmat = [], []
    for u in range(a.utts):
        s, t, _ = synth_alignment_task(sseq, pssid, ptmat, m.n_ciphone, a.phones, 777 + u)
        senid.append(s)
        tmat.append(t)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    frame_off = (np.arange(a.utts + 1) * a.frames).astype(np.int32)
    phone_off = (np.arange(a.utts + 1) * a.phones).astype(np.int32)
    d_feats = m.to_device(feats)
    d_scr = L.ssw_device_malloc(C.c_size_t(n_total * m.n_sen * 2))
    best = {"score_s": 1e9, "align_s": 1e9}
    for _ in range(a.reps):
        t0 = time.perf_counter()
        m.score_batch_device(d_feats, n_total, frame_off, d_scr)
        L.ssw_device_synchronize()
        t1 = time.perf_counter()
        st, status = m.align_batch(d_scr, frame_off, phone_off, senid, tmat)
        t2 = time.perf_counter()
        best["score_s"] = min(best["score_s"], t1 - t0)
        best["align_s"] = min(best["align_s"], t2 - t1)
    ok = int((status == 0).sum())
    wall = best["score_s"] + best["align_s"]
    print(json.dumps({
        "workload": f"{a.utts} utterances x {a.frames} frames x {a.phones} phones, en-us",
        "score_s": best["score_s"], "align_s": best["align_s"],
        "score_frames_per_s": n_total / best["score_s"],
        "align_utt_frames_per_s": n_total / best["align_s"],
        "align_rtf": wall / (n_total / 100.0),
        "aligned_ok": ok, "n_utts": a.utts,
        "tiles_ok": bool(all(
            (st[phone_off[u] * 3:phone_off[u + 1] * 3, 1].sum() == a.frames) for u in range(a.utts)
            if status[u] == 0)),
    }))


if __name__ == "__main__":
    main()
