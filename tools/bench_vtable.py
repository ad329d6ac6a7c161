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
    out = {
        "workload": "en-us PTM, 1000 frames through vt->frame_eval (compallsen=yes), ctypes caller",
        "frame_eval_us": per_frame * 1e6, "frames_per_s": 1.0 / per_frame,
        "prescore_ms_for_1000_frames": t_pre * 1e3,
        "frame_eval_after_prescore_us": per_row * 1e6,
        "note": "the ctypes wrapper costs ~40 us per call (frame_eval_after_prescore_us is a "
                "memcpy in C): `from_c` is the same loop from a C caller"}
    g.free()
    m = None
    # the same from C (tools/vtable_latency.c), both configurations
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.mkdtemp(), "vtable_latency")
    lib = os.path.join(root, "soundswallower_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tools", "vtable_latency.c"), "-L", lib, "-lssw_amd",
                           "-lm", "-Wl,-rpath," + lib, "-o", exe])
    import torch                                  # its ROCm libraries are the ones to load
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + \
        env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, mdir, "1000"], capture_output=True, text=True, env=env)
    out["from_c"] = json.loads(r.stdout) if r.returncode == 0 else {"error": r.stderr[-300:]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
