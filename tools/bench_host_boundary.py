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
        n = len(feats)
        rows = np.zeros((n, m.n_sen), np.int16)       # the caller's buffer, reused call after call
        for _ in range(3):
            m.score_batch(feats, off, out=rows)
        reps = 20 if utts == 16 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            m.score_batch(feats, off, out=rows)
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            m.score_batch(feats, off)                 # a fresh (zeroed, untouched) buffer per call
        dt_fresh = (time.perf_counter() - t0) / reps
        # a ragged stream of batches: every call other utterance offsets (the uploads of the
        # offsets and start bits), against the same frames with fixed offsets
        rng = np.random.default_rng(utts)
        offs = []
        for _ in range(reps):
            cuts = np.sort(rng.choice(np.arange(1, n), size=utts - 1, replace=False))
            offs.append(np.concatenate([[0], cuts, [n]]).astype(np.int32))
        t0 = time.perf_counter()
        for o in offs:
            m.score_batch(feats, o, out=rows)
        dt_ragged = (time.perf_counter() - t0) / reps
        out[f"frames_{n}"] = {"ms_per_call": dt * 1e3, "frames_per_s": n / dt,
                              "frames_per_s_fresh_output_buffer": n / dt_fresh,
                              "frames_per_s_new_offsets_every_call": n / dt_ragged,
                              "host_bytes_per_call": int(feats.nbytes + n * m.n_sen * 2),
                              "effective_GBps": (feats.nbytes + n * m.n_sen * 2) / dt / 1e9}
    # the device-resident call on a ragged stream: new offsets every call against fixed ones
    import ctypes as C
    feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(16)])
    n = len(feats)
    d_feats = m.to_device(feats)
    d_out = m.device_malloc(n * m.n_sen * 2)
    rng = np.random.default_rng(7)
    offs = []
    for _ in range(200):
        cuts = np.sort(rng.choice(np.arange(1, n), size=15, replace=False))
        offs.append(np.concatenate([[0], cuts, [n]]).astype(np.int32))
    fixed = (np.arange(17) * 256).astype(np.int32)
    res = {}
    for name, seq in (("fixed_offsets", [fixed] * 200), ("new_offsets_every_call", offs)):
        for o in seq[:20]:
            m.score_batch_device(d_feats, n, o, d_out)
        m._L.ssw_device_synchronize()
        t0 = time.perf_counter()
        for o in seq:
            m.score_batch_device(d_feats, n, o, d_out)
        m._L.ssw_device_synchronize()
        res[name] = n * len(seq) / (time.perf_counter() - t0)
    m.device_free(d_feats)
    m.device_free(d_out)
    out["device_resident_4096_frames"] = dict(res, ragged_over_fixed=res["new_offsets_every_call"]
                                              / res["fixed_offsets"])
    out["note"] = ("ssw_score_batch_host: host buffers; round 5: sub-batches of ~1024 frames "
                   "through pinned staging, scoring / download / host copy overlapped (10,252 B "
                   "of scores per frame against 156 B of features); device_resident: "
                   "ssw_score_batch, 4096 frames per call, through the Python binding")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
