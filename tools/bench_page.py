#!/usr/bin/env python3
"""A ReadAlongs-sized unit of work: ONE 2,000-word page (about 19 K phone-tree HMMs, 8 K
word-final ones, 13 K phones, ~96 K frames = 16 minutes of audio) through decoder_alignment on
the GPU (ssw_forced_align_batch: first pass over the text's grammar, alignment_populate with its
word windows, constrained state alignment, propagate), senone scores already in HBM.  Synthetic
scores that follow one path through the text (tests/test_gpu_first_pass.py: synth_scores).
Prints one JSON line: seconds per call (first call = workspaces allocated, then steady state),
the stage split of the last call (SSW_ALIGN_TIMING) and device memory in use."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import soundswallower_amd as ssw
    from oracle import fsg_oracle as F        # synthetic input only (not the measured path)
    from oracle import oracle as O
    from soundswallower_amd.synth import lcg_uniform
    from tests.test_gpu_first_pass import synth_scores
    n_words = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    orc = O.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    olex = F.Lexicon(orc, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    words = [vocab[int(x * len(vocab))] for x in lcg_uniform(11, n_words)]
    scr = synth_scores(F, orc, olex, words, 5, orc.n_sen, sil_p=0.1)
    off = np.array([0, len(scr)], np.int32)
    d = torch.from_numpy(np.ascontiguousarray(scr, np.int16)).cuda()
    free0, total = torch.cuda.mem_get_info()
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        aset = ssw.forced_align_batch(m, lex, d, off, [words])
        times.append(time.perf_counter() - t0)
        ok = aset.status(0) == 0
        n_phones = len(aset.utterance(0)["cipid"]) if ok else 0
        aset.free()
    free1, _ = torch.cuda.mem_get_info()
    print(json.dumps({
        "workload": f"one page of {n_words} words, {len(scr)} frames ({len(scr) / 6000:.1f} min of "
                    f"audio), {n_phones} phones: ssw_forced_align_batch, scores resident in HBM",
        "aligned": bool(ok), "first_call_s": times[0], "steady_s": min(times[1:]),
        "rtf": min(times[1:]) / (len(scr) / 100.0),
        "score_array_GB": scr.nbytes / 1e9,
        "workspace_GB": (free0 - free1) / 1e9}))


if __name__ == "__main__":
    main()
