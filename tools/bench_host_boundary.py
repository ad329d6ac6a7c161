#!/usr/bin/env python3
"""The PCIe-inclusive rate of the scoring path (run ON the GPU box): ssw_score_batch_host takes
HOST buffers -- features in, int16 score rows out -- as the reference's acmod_score does.  The bench
line's `value` is measured with inputs and outputs resident in HBM (ssw_score_batch); this is
what a caller that keeps nothing on the device sees.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def main():
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    means = read_raw_means(mdir)
    out = {}
    for utts in (16, 256):
        feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(utts)])
        off = (np.arange(utts + 1) * 256).astype(np.int32)
        for _ in range(3):
            m.score_batch(feats, off)
        reps = 20 if utts == 16 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            m.score_batch(feats, off)
        dt = (time.perf_counter() - t0) / reps
        n = len(feats)
        out[f"frames_{n}"] = {"ms_per_call": dt * 1e3, "frames_per_s": n / dt,
                              "host_bytes_per_call": int(feats.nbytes + n * m.n_sen * 2),
                              "effective_GBps": (feats.nbytes + n * m.n_sen * 2) / dt / 1e9}
    out["note"] = ("ssw_score_batch_host: pageable host buffers, hipMemcpy in, two kernels, "
                   "hipMemcpy out (10,252 B of scores per frame against 156 B of features)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
