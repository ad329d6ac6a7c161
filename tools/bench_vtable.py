#!/usr/bin/env python3
"""Latency of the drop-in vtable call: ssw_mgau_t.vt->frame_eval one frame at a time (what
acmod_score does through the unchanged reference decoder), with and without ssw_mgau_prescore."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def main():
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    feats = synth_features(read_raw_means(mdir), 1000, 4711)
    g = ssw.PtmMgau(m)
    for f in range(50):
        g.frame_eval(feats[f], f)
    g.reset_hist()
    t0 = time.perf_counter()
    for f in range(len(feats)):
        g.frame_eval(feats[f], f)
    per_frame = (time.perf_counter() - t0) / len(feats)
    g.reset_hist()
    t0 = time.perf_counter()
    g.prescore(feats)
    t_pre = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f in range(len(feats)):
        g.frame_eval(feats[f], f)
    per_row = (time.perf_counter() - t0) / len(feats)
    print(json.dumps({
        "workload": "en-us PTM, 1000 frames through vt->frame_eval (compallsen=yes), ctypes caller",
        "frame_eval_us": per_frame * 1e6, "frames_per_s": 1.0 / per_frame,
        "prescore_ms_for_1000_frames": t_pre * 1e3,
        "frame_eval_after_prescore_us": per_row * 1e6}))


if __name__ == "__main__":
    main()
