#!/usr/bin/env python3
"""What a frame-synchronous batched first pass would pay for scoring alone.

In the reference's default configuration (compallsen = no) frame t of an utterance can only be
scored once its frame t - 1 has been searched: the active senone set is the beam search's state
(src/fsg_search.c:316, 664-739; src/acmod.c:905-999).  The only batch dimension left is the
utterances, so a batched first pass in that configuration is a loop over frames that scores ONE
frame of every utterance per step.  This tool times that loop's scoring half in the most
favourable form -- every senone (no list building), no search step, launched back to back from C
(ssw_debug_score_loop) -- for 256 utterances x 1000 frames, next to the same 256,000 frames as one
batch.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def main():
    n_utts, n_frames = 256, 1000
    L = _lib.lib()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    means = read_raw_means(mdir)
    step = synth_features(means, n_utts, 77)               # one frame of every utterance
    off1 = np.arange(n_utts + 1, dtype=np.int32)
    d_step = m.to_device(step)
    d_out1 = m.device_malloc(n_utts * m.n_sen * 2)
    rv = L.ssw_debug_score_loop(m._m, 0, d_step, n_utts, off1.ctypes.data, n_utts, d_out1, 50, None)
    assert rv == 0
    t0 = time.perf_counter()
    rv = L.ssw_debug_score_loop(m._m, 0, d_step, n_utts, off1.ctypes.data, n_utts, d_out1, n_frames, None)
    loop_s = time.perf_counter() - t0
    assert rv == 0
    full = np.concatenate([synth_features(means, n_frames, 12345 + u) for u in range(8)])
    full = np.tile(full, (n_utts // 8, 1))
    offb = (np.arange(n_utts + 1) * n_frames).astype(np.int32)
    d_full = m.to_device(full)
    d_outb = m.device_malloc(n_utts * n_frames * m.n_sen * 2)
    L.ssw_debug_score_loop(m._m, 0, d_full, n_utts * n_frames, offb.ctypes.data, n_utts, d_outb, 1, None)
    t0 = time.perf_counter()
    L.ssw_debug_score_loop(m._m, 0, d_full, n_utts * n_frames, offb.ctypes.data, n_utts, d_outb, 3, None)
    batch_s = (time.perf_counter() - t0) / 3
    print(json.dumps({
        "workload": f"{n_utts} utterances x {n_frames} frames, en-us PTM, every senone",
        "frame_synchronous_loop_ms": loop_s * 1e3,
        "per_step_us": loop_s / n_frames * 1e6,
        "one_batch_ms": batch_s * 1e3,
        "ratio": loop_s / batch_s,
        "note": "the loop scores one frame of every utterance per step (2 launches per step), "
                "which is what a batched first pass in the default configuration has to do before "
                "its search step; the search step, the active-list construction and the "
                "history hand-over per utterance come on top"}))


if __name__ == "__main__":
    main()
