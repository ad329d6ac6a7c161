#!/usr/bin/env python3
"""Experiment (ON the GPU box): where do the first steps of a short timed region go?  The driver
times 20 steps after 5 warm-up steps and a synchronisation; the region's own events read ~4 us
per step more than a 200-step region.  Per-step event stamps of such a region, several times."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def main():
    mdir = ssw.model_dir("en-us")
    means = read_raw_means(mdir)
    feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(16)])
    off = (np.arange(17) * 256).astype(np.int32)
    torch.cuda.init()
    m = ssw.Model(mdir)
    step = bench.ScoreStep(torch, m, feats, off)
    bench.spin_up(torch, step, 0.5)
    out = {}
    for name, idle_s in (("after_sync", 0.0), ("after_1ms_idle", 0.001), ("after_20ms_idle", 0.02),
                         ("after_sync_again", 0.0)):
        rows = []
        for rep in range(5):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            if idle_s:
                time.sleep(idle_s)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            t0 = time.perf_counter()
            ev[0].record()
            host = []
            for i in range(20):
                step()
                ev[i + 1].record()
                host.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e6
            per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20)]
            rows.append({"per_step_us": [round(x, 1) for x in per], "region_us": round(sum(per), 1),
                         "wall_us": round(wall, 1),
                         "host_launch_done_us": [round(host[0], 1), round(host[1], 1), round(host[-1], 1)]})
        out[name] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
