#!/usr/bin/env python3
"""BASELINE config 4 timing: the ms scorer (ms_gauden + ms_senone) on fr-fr, 8192-frame batch.

The shipped fr-fr model carries a sendump only, so the `mixture_weights` file the ms scorer
reads is synthesised from it (pdf = 1.0001^-(q*1024)), exactly as the parity
tests do.  Prints one JSON line."""
import argparse
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def write_s3(path, payload):
    hdr = b"s3\nversion 1.0\nchksum0 yes\nendhdr\n"
    s = 0
    for w in np.frombuffer(payload, "<u4").tolist():
        s = (((s << 20) | (s >> 12)) + w) & 0xFFFFFFFF
    with open(path, "wb") as fh:
        fh.write(hdr + struct.pack("<I", 0x11223344) + payload + struct.pack("<I", s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=32)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    a = ap.parse_args()
    _lib.build()
    L = _lib.lib()
    mdir = ssw.model_dir("fr-fr")
    tables = ssw.Model(mdir, config={"device": -2})
    q = tables.table("ptm_mixw").reshape(tables.n_feat, tables.n_density, tables.n_sen)
    pdf = np.ascontiguousarray(np.power(1.0001, -(q.astype(np.float64) * 1024.0)).transpose(2, 0, 1),
                               dtype="<f4")
    tmp = tempfile.mkdtemp()
    mixw = os.path.join(tmp, "mixture_weights")
    write_s3(mixw, struct.pack("<4i", pdf.shape[0], pdf.shape[1], pdf.shape[2], pdf.size)
             + pdf.tobytes())
    m = ssw.Model(mdef=os.path.join(mdir, "mdef"), means=os.path.join(mdir, "means"),
                  variances=os.path.join(mdir, "variances"),
                  tmat=os.path.join(mdir, "transition_matrices"), mixw=mixw)
    means = read_raw_means(mdir)
    feats = np.concatenate([synth_features(means, a.frames, 4242 + u) for u in range(a.utts)])
    n = feats.shape[0]
    off = (np.arange(a.utts + 1) * a.frames).astype(np.int32)
    d_feats = m.to_device(feats)
    d_out = L.ssw_device_malloc(n * m.n_sen * 2)

    def step():
        m.score_batch_device(d_feats, n, off, d_out, scorer=ssw.SCORER_MS)

    for _ in range(a.warmup):
        step()
    L.ssw_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    L.ssw_device_synchronize()
    dt = time.perf_counter() - t0
    m.set_kernel_timing(True)
    k = np.zeros(2)
    for _ in range(20):
        step()
        k += np.array(m.kernel_timing())
    k /= 20
    flagged, pairs = m.last_stats()
    print(json.dumps({
        "workload": f"ms scorer (ms_gauden + ms_senone), fr-fr, {a.utts} x {a.frames} = {n} frames per step",
        "frames_per_s": n * a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
        "topn_kernels_ms": float(k[0]), "senone_kernel_ms": float(k[1]),
        "exact_pass_share": flagged / max(pairs, 1),
        "n_sen": m.n_sen, "n_cb": m.n_cb, "n_density": m.n_density}))


if __name__ == "__main__":
    main()
