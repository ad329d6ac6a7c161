#!/usr/bin/env python3
"""Experiment (ON the GPU box): does using a second HIP stream in the process slow later launches
on the first one down?  tools/exp_two_streams.py saw one stream run at 2 x the time per 4096-frame
batch AFTER a phase that alternated two streams.  Here, one model: the headline step on the null
stream / on a created stream, before and after (a) ssw_score_batch_host (two internal streams),
(b) a phase that alternates two created streams, (c) an idle second stream merely created."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def main():
    mdir = ssw.model_dir("en-us")
    means = read_raw_means(mdir)
    feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(16)])
    off = (np.arange(17) * 256).astype(np.int32)
    n = len(feats)
    torch.cuda.init()
    m = ssw.Model(mdir)
    d_feats = m.to_device(feats)
    d_out = m.device_malloc(n * m.n_sen * 2)
    d_out2 = m.device_malloc(n * m.n_sen * 2)
    res = {}

    def run(name, streams, steps=600):
        for i in range(100):
            m.score_batch_device(d_feats, n, off, d_out, stream=streams[i % len(streams)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            m.score_batch_device(d_feats, n, off, d_out if i % 2 == 0 else d_out2,
                                 stream=streams[i % len(streams)])
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / steps * 1e6, 1)

    run("null_stream_first", [None])
    run("null_stream_again", [None])
    s1 = torch.cuda.Stream()
    run("created_stream", [s1.cuda_stream])
    run("null_stream_after_created", [None])
    s2 = torch.cuda.Stream()
    run("created_stream_with_idle_second", [s1.cuda_stream])
    run("two_streams_alternating", [s1.cuda_stream, s2.cuda_stream])
    run("created_stream_after_two", [s1.cuda_stream])
    run("null_stream_after_two", [None])
    rows = np.zeros((n, m.n_sen), np.int16)
    m.score_batch(feats, off, out=rows)                 # ssw_score_batch_host: internal streams
    run("null_stream_after_host_call", [None])
    run("created_stream_after_host_call", [s1.cuda_stream])
    time.sleep(2.0)
    run("null_stream_after_2s_idle", [None])
    print(json.dumps({"us_per_4096_frame_batch": res}))


if __name__ == "__main__":
    main()
