#!/usr/bin/env python3
"""The reference's DEFAULT configuration (compallsen = no) from text, as a batch, timed on one
MI355X: ssw_first_pass_batch_active (the first pass by speculation and proof) and
ssw_align_text_batch_active (decoder_alignment: that first pass, populate, the second pass over
the growing active set, propagate), next to the compallsen = yes calls on the same workload
(tools/bench_first_pass.py: N utterances x F frames, texts of W words, synthetic audio that
follows the text) and, optionally, on the reference's recording tiled N times with its text.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib  # noqa: E402
from tools.bench_first_pass import build_workload  # noqa: E402


def timed(fn, reps, torch):
    best, out = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def run(ssw, m, lex, torch, utts=256, frames=1000, words_per_text=25, reps=3, noise=0.3,
        texts=None, feats=None):
    if texts is None:
        texts, feats, nodes_per_text = build_workload(ssw, m, lex, utts, frames, words_per_text, noise)
    else:
        nodes_per_text = float(np.mean([len(lex.first_pass_graph(t)[0]) for t in texts[:8]]))
    off = (np.arange(utts + 1) * frames).astype(np.int32)
    d_feats = torch.from_numpy(feats).cuda()
    d_scr = torch.empty((len(feats), m.n_sen), dtype=torch.int16, device="cuda")
    marshalled = ssw.Texts(texts)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        m.score_batch_device(d_feats, len(feats), off, d_scr)
        torch.cuda.synchronize()

    def yes_first():
        m.score_batch_device(d_feats, len(feats), off, d_scr)
        return lex.first_pass_raw(d_scr, off, texts)

    def yes_all():
        a = ssw.align_text_batch(m, lex, d_feats, off, marshalled)
        st = [a.status(k) for k in range(utts)]
        a.free()
        return st

    def no_all():
        a = ssw.align_text_batch_active(m, lex, d_feats, off, marshalled)
        st = [a.status(k) for k in range(utts)]
        a.free()
        return st

    s0 = m.first_pass_active_stats()
    fp_no_ms, (_, _, rounds, _) = timed(lambda: lex.first_pass_active_raw(d_feats, off, marshalled),
                                        reps, torch)
    s1 = m.first_pass_active_stats()
    segs_no, _ = lex.first_pass_active(d_feats, off, texts)
    fp_yes_ms, _ = timed(yes_first, reps, torch)
    all_yes_ms, st_yes = timed(yes_all, reps, torch)
    all_no_ms, st_no = timed(no_all, reps, torch)
    m.score_batch_device(d_feats, len(feats), off, d_scr)
    segs_yes = lex.first_pass(d_scr, off, texts)
    same = sum(1 for a, b in zip(segs_yes, segs_no)
               if (a is None) == (b is None)
               and (a is None or [(w, s, d) for (w, s, d, _) in a] == [(w, s, d) for (w, s, d, _) in b]))
    hist = np.bincount(rounds, minlength=2).tolist()
    audio_s = utts * frames / 100.0
    return {
        "workload": f"{utts} utterances x {frames} frames, texts of {len(texts[0])} words, en-us: the "
                    f"reference's default configuration (compallsen = no) from features + text",
        "hmms_per_text": nodes_per_text,
        "first_pass_active_ms": fp_no_ms,
        "first_pass_yes_ms_scoring_included": fp_yes_ms,
        "decoder_alignment_active_ms": all_no_ms,
        "decoder_alignment_yes_ms": all_yes_ms,
        "rtf_active": all_no_ms / 1e3 / audio_s,
        "rounds_histogram": hist,
        "rounds_of_the_call": int((s1[2])),
        "rounds_mean": float(rounds.mean()),
        "first_pass_completed": sum(s is not None for s in segs_no),
        "aligned_active": sum(s == 0 for s in st_no), "aligned_yes": sum(s == 0 for s in st_yes),
        "word_boundaries_equal_to_compallsen_yes": same, "n_utts": utts,
        "per_frame_path_estimate_ms": utts * frames * 0.023,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--words", type=int, default=25)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--noise", type=float, default=0.3)
    ap.add_argument("--real", action="store_true",
                    help="the reference's recording (tests/golden/goforward_mfcc.npy) tiled "
                         "--utts times with its text instead of synthetic audio")
    a = ap.parse_args()
    import torch
    _lib.build()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    if a.real:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cep = np.load(os.path.join(root, "tests", "golden", "goforward_mfcc.npy")).astype(np.float32)
        n = len(cep)
        off = (np.arange(a.utts + 1) * n).astype(np.int32)
        feats = m.feat_batch(np.tile(cep, (a.utts, 1)), utt_off=off)
        pool = ["go forward ten meters", "go forward", "forward ten", "hello world", "ten",
                "go ten meters forward"]
        texts = [pool[k % len(pool)].split() for k in range(a.utts)]
        out = run(ssw, m, lex, torch, a.utts, n, 0, a.reps, a.noise, texts=texts, feats=feats)
        out["workload"] = (f"{a.utts} x the reference's recording goforward (278 frames) with six "
                           f"texts in turn, en-us: default configuration from features + text")
    else:
        out = run(ssw, m, lex, torch, a.utts, a.frames, a.words, a.reps, a.noise)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
