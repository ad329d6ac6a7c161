#!/usr/bin/env python3
"""Experiment (run ON the GPU box): do the scan (matrix cores + vector unit, 56 % vector-busy) and
the senone kernel (70 % vector-busy) of DIFFERENT batches share the chip well?  Two models on two
streams score the headline batch alternately, against one model on one stream.  If the pair is
clearly faster per batch than the single stream, a scoring pipeline that overlaps batch k + 1's
scan with batch k's senone kernel is worth building into the library.

    python tools/exp_two_streams.py [--steps 400] [--frames 4096]
"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--models", type=int, default=2)
    a = ap.parse_args()
    mdir = ssw.model_dir("en-us")
    means = read_raw_means(mdir)
    utts = a.frames // 256
    feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(utts)])
    off = (np.arange(utts + 1) * 256).astype(np.int32)
    n = len(feats)
    torch.cuda.init()
    ms = [ssw.Model(mdir) for _ in range(a.models)]
    st = [torch.cuda.Stream() for _ in range(a.models)]
    d_feats = ms[0].to_device(feats)
    d_out = [m.device_malloc(n * m.n_sen * 2) for m in ms]
    res = {}
    for name, k in (("one_stream", 1), (f"{a.models}_streams", a.models), ("one_stream_again", 1)):
        for i in range(40):
            ms[i % k].score_batch_device(d_feats, n, off, d_out[i % k], stream=st[i % k].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            ms[i % k].score_batch_device(d_feats, n, off, d_out[i % k], stream=st[i % k].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"us_per_batch": dt / a.steps * 1e6, "frames_per_s": n * a.steps / dt}
    res["knobs"] = {k: v for k, v in os.environ.items() if k.startswith("SSW_")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
