#!/usr/bin/env python3
"""Experiment (ON the GPU box): long texts on REAL speech.  The measurements behind "texts beyond
1,024 phone-tree HMMs take the sliding-window kernel" used synthetic audio that follows its text
closely; on real audio the first pass's active range is wider, and an utterance that outgrows the
window is searched again one level down.  Here: the reference's own recording (goforward, 278
frames, "go forward ten meters") repeated R times as one utterance with its text repeated R
times -- 8 such utterances per batch -- through ssw_align_text_batch, with the default kernel
choice, with the window kernel switched off (SSW_FP_WIN=0: HBM-resident kernel at once) and with
a wider window (SSW_FP_WIN_TPB=1024).  Same alignments in every mode; times per batch."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SSW_KNOBS_DYNAMIC", "1")
import torch  # noqa: E402
import soundswallower_amd as ssw  # noqa: E402


def main():
    m = ssw.Model(ssw.model_dir("en-us"))
    lex = ssw.Lexicon(m, ssw.model_dir("en-us") + "/dict.txt", ssw.model_dir("en-us") + "/noisedict.txt")
    cep = np.load(os.path.join(ROOT, "tests", "golden", "goforward_mfcc.npy")).astype(np.float32)
    out = {}
    for reps in (30, 75, 150):
        one = np.concatenate([cep] * reps)
        n_utts = 8
        ceps = np.concatenate([one] * n_utts)
        off = (np.arange(n_utts + 1) * len(one)).astype(np.int32)
        feats = m.feat_batch(ceps, off)
        d = torch.from_numpy(feats).cuda()
        text = ["go", "forward", "ten", "meters"] * reps
        texts = [text] * n_utts
        n_nodes = len(lex.first_pass_graph(text)[0])
        row = {"words": len(text), "frames": len(one), "phone_tree_hmms": n_nodes}
        ref = None
        for name, env in (("default", {}), ("window_off", {"SSW_FP_WIN": "0"}),
                          ("window_1024", {"SSW_FP_WIN_TPB": "1024"})):
            os.environ.update(env)
            try:
                best = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    res = ssw.align_text_batch(m, lex, d, off, texts)
                    dt = (time.perf_counter() - t0) * 1e3
                    u0 = res.utterance(0)
                    if u0 is not None:
                        u0 = {"words": list(u0["words"]), "word_al": np.array(u0["word_al"]).copy()}
                    res.free()
                    best = dt if best is None else min(best, dt)
                words = None if u0 is None else (list(u0["words"]), np.array(u0["word_al"]).tolist())
                row[name + "_ms"] = round(best, 2)
                row[name + "_aligned"] = u0 is not None
                if ref is None:
                    ref = words
                row[name + "_same_words"] = words == ref
            finally:
                for k in env:
                    del os.environ[k]
        out[f"x{reps}"] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
