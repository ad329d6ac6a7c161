#!/usr/bin/env python3
"""First pass + second pass of forced alignment from TEXT, timed on one MI355X.

N utterances x F frames; every utterance is a text of W random dictionary words.  The audio is
synthetic but follows the text: for every HMM state along one path through the text's phone
trees the frame's features sit on the mean of the best-weighted Gaussian of that state's senone
(per stream), plus noise -- so the first pass has something to find, as it would on speech.
Timed: senone scoring; the first pass alone (ssw_first_pass_batch: graphs built on the host,
search on the GPU); and decoder_alignment as a whole (first pass + populate + constrained state
alignment + propagate, soundswallower_amd.forced_alignment).
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
from soundswallower_amd.synth import lcg_uniform, read_raw_means  # noqa: E402


def path_through(lex, nodes, n_words, rng):
    """one node sequence <sil> w1 ... wN <sil> through the graph of ssw_first_pass_graph"""
    kids = {}
    for i, n in enumerate(nodes):
        if n["parent"] >= 0:
            kids.setdefault(int(n["parent"]), []).append(i)
    filler = [bool(n["flags"] & 2) and lex.word(int(n["wid"])).startswith(("<", "["))
              for n in nodes]
    def sil(state):
        for i, n in enumerate(nodes):
            if filler[i] and n["state"] == state and lex.word(int(n["wid"])) == "<sil>":
                return i
    path = [sil(0)]
    for s in range(n_words):
        roots = [i for i, n in enumerate(nodes)
                 if (n["flags"] & 1) and n["state"] == s and not filler[i]]
        i = roots[rng.integers(len(roots))]
        while True:
            path.append(i)
            if nodes[i]["flags"] & 2:
                break
            i = kids[i][rng.integers(len(kids[i]))]
    path.append(sil(n_words))
    return path


def build_workload(ssw, m, lex, utts, frames, words_per_text, noise=0.3, seed=4242):
    """texts + features [utts * frames][39] that follow them (see the module docstring)"""
    mdir = ssw.model_dir("en-us")
    means = read_raw_means(mdir)                                  # [cb][feat][density][13]
    mixw = m.table("ptm_mixw").reshape(m.n_feat, m.n_density, m.n_sen)
    sen2cb = m.table("sen2cb")
    best_d = mixw.argmin(axis=1)                                  # [feat][sen]
    sen_mean = np.concatenate([means[sen2cb, f, best_d[f]] for f in range(m.n_feat)], axis=1)
    n_dict = len(lex)
    u = lcg_uniform(seed, 4 * utts * words_per_text)
    k = 0
    texts, feats, n_nodes = [], [], []
    rng = np.random.default_rng(1)
    for t in range(utts):
        while True:
            words = []
            while len(words) < words_per_text:
                w = lex.word(int(u[k % len(u)] * n_dict))
                k += 1
                if w and "(" not in w and not w.startswith(("<", "[")):
                    words.append(w)
            nodes, _ = lex.first_pass_graph(words)
            path = path_through(lex, nodes, words_per_text, rng)
            if 3 * len(path) <= frames:
                break
        states = np.array([s for i in path for s in nodes[i]["senid"]])
        per = np.full(len(states), frames // len(states))
        per[-3:] += (frames - per.sum() + 2) // 3
        per[-1] += frames - per.sum()
        sen = np.repeat(states, per)
        x = sen_mean[sen] + rng.standard_normal((frames, sen_mean.shape[1])).astype(np.float32) * noise
        texts.append(words)
        feats.append(x.astype(np.float32))
        n_nodes.append(len(nodes))
    return texts, np.concatenate(feats), float(np.mean(n_nodes))


def run(ssw, m, lex, torch, utts=256, frames=1000, words_per_text=25, reps=3, noise=0.3):
    texts, feats, nodes_per_text = build_workload(ssw, m, lex, utts, frames, words_per_text, noise)
    off = (np.arange(utts + 1) * frames).astype(np.int32)
    d_feats = torch.from_numpy(feats).cuda()
    d_scr = torch.empty((len(feats), m.n_sen), dtype=torch.int16, device="cuda")
    best = None
    marshalled = ssw.Texts(texts)      # char ** + offsets, built once (a C host has them anyway)
    t_spin = time.perf_counter()       # the GPU idled while the host made the workload
    while time.perf_counter() - t_spin < 0.3:
        m.score_batch_device(d_feats, len(feats), off, d_scr)
        torch.cuda.synchronize()
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.score_batch_device(d_feats, len(feats), off, d_scr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lex.first_pass_raw(d_scr, off, texts)
        t2 = time.perf_counter()
        aset = ssw.forced_align_batch(m, lex, d_scr, off, texts)   # first pass again + the rest
        t3 = time.perf_counter()
        res = [aset.utterance(k) for k in range(utts)]
        aset.free()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        one = ssw.align_text_batch(m, lex, d_feats, off, marshalled)   # everything in one call: the
        t5 = time.perf_counter()                                   # graphs are built while the GPU scores
        one.free()
        cur = {"score_s": t1 - t0, "first_pass_s": t2 - t1, "alignment_s": t3 - t2, "one_call_s": t5 - t4}
        best = cur if best is None else {k: min(best[k], cur[k]) for k in cur}   # per stage
    segs = lex.first_pass(d_scr, off, texts)
    done = [s for s in segs if s is not None]
    same_words = sum(1 for s, t in zip(segs, texts) if s is not None and
                     [w.split("(")[0] for (w, _, _, _) in s if not w.startswith(("<", "["))] == t)
    aligned = sum(1 for r in res if r is not None)
    tiles = all(int(r["word_al"][:, 1].sum()) == frames for r in res if r is not None)
    audio_s = utts * frames / 100.0
    wall = best["one_call_s"]
    return {
        "workload": f"{utts} utterances x {frames} frames, texts of {words_per_text} words, en-us; "
                    f"synthetic audio following the text: scoring, then decoder_alignment from "
                    f"text (first pass + populate + constrained state alignment)",
        "hmms_per_text": nodes_per_text,
        "score_ms": best["score_s"] * 1e3, "first_pass_ms": best["first_pass_s"] * 1e3,
        "decoder_alignment_ms": best["alignment_s"] * 1e3,
        "features_to_alignments_ms": best["one_call_s"] * 1e3,
        "first_pass_completed": len(done), "first_pass_words_equal_text": same_words,
        "aligned": aligned, "alignments_tile_their_utterances": bool(tiles), "n_utts": utts,
        "rtf": wall / audio_s,
        "first_pass_utt_frames_per_s": utts * frames / best["first_pass_s"],
    }


def run_pipeline(ssw, m, lex, torch, utts, frames, words_per_text, batches, noise=0.3):
    """Steady state of a stream of batches: a host thread prepares the next batch's graphs
    (ssw_first_pass_prepare: no device call, the GIL is released inside the C call) while the
    GPU scores and searches the current one.  Returns ms per batch, pipelined and not."""
    import threading
    texts, feats, _ = build_workload(ssw, m, lex, utts, frames, words_per_text, noise)
    off = (np.arange(utts + 1) * frames).astype(np.int32)
    d_feats = torch.from_numpy(feats).cuda()
    d_scr = torch.empty((len(feats), m.n_sen), dtype=torch.int16, device="cuda")

    def gpu_part(plan):
        m.score_batch_device(d_feats, len(feats), off, d_scr)
        aset = ssw.forced_align_planned(m, lex, plan, d_scr, off)
        n = sum(aset.status(k) == 0 for k in range(utts))
        aset.free()
        return n

    # warm-up, then the two ways
    p0 = ssw.FirstPassPlan(m, lex, texts)
    gpu_part(p0)
    p0.free()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batches):
        plan = ssw.FirstPassPlan(m, lex, texts)
        ok = gpu_part(plan)
        plan.free()
    serial = (time.perf_counter() - t0) / batches
    box = {}

    def prep():
        box["plan"] = ssw.FirstPassPlan(m, lex, texts)

    prep()
    t0 = time.perf_counter()
    for _ in range(batches):
        cur = box["plan"]
        th = threading.Thread(target=prep)
        th.start()
        ok = gpu_part(cur)
        th.join()
        cur.free()
    piped = (time.perf_counter() - t0) / batches
    box["plan"].free()
    audio_s = utts * frames / 100.0
    return {"workload": f"stream of batches of {utts} utterances x {frames} frames, texts of "
                        f"{words_per_text} words: features + text -> alignments",
            "batches": batches, "aligned_in_last_batch": ok,
            "ms_per_batch_serial": serial * 1e3, "ms_per_batch_pipelined": piped * 1e3,
            "rtf_serial": serial / audio_s, "rtf_pipelined": piped / audio_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pipeline", type=int, default=0,
                    help="N > 0: stream N batches, the next batch's graphs prepared on a host "
                         "thread while the GPU works on the current one")
    ap.add_argument("--utts", type=int, default=256, help="utterances in the whole job")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--words", type=int, default=25)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--noise", type=float, default=0.3)
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for the gather (nccl = RCCL; gloo lets two "
                         "ranks share one GPU on a 1-GPU box to exercise the multi-rank flow)")
    ap.add_argument("--device", type=int, default=None, help="GPU to use (default LOCAL_RANK)")
    a = ap.parse_args()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if a.device is None else a.device
    torch.cuda.set_device(local_rank)
    if world == 1:
        _lib.build()
        mdir = ssw.model_dir("en-us")
        m = ssw.Model(mdir)
        lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
        if a.pipeline > 0:
            print(json.dumps(run_pipeline(ssw, m, lex, torch, a.utts, a.frames, a.words, a.pipeline,
                                          a.noise)))
        else:
            print(json.dumps(run(ssw, m, lex, torch, a.utts, a.frames, a.words, a.reps, a.noise)))
        return
    # launched as `python -m torch.distributed.run --nproc-per-node N tools/bench_first_pass.py`:
    # the job's texts are dealt to the ranks, every rank scores and aligns its own, and the
    # alignments (their sizes depend on the fillers and alternates found) are gathered once
    import torch.distributed as dist
    from soundswallower_amd.parallel import gather_text_alignments, shard_utterances
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(a.backend)
    if rank == 0:
        _lib.build()
    dist.barrier()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir, config={"device": local_rank})
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    texts, feats, _ = build_workload(ssw, m, lex, a.utts, a.frames, a.words, a.noise)  # same on all ranks
    mine = shard_utterances([a.frames] * a.utts, world)[rank]
    my_texts = [texts[u] for u in mine]
    my_feats = np.concatenate([feats[u * a.frames:(u + 1) * a.frames] for u in mine])
    off = (np.arange(len(mine) + 1) * a.frames).astype(np.int32)
    d_feats = torch.from_numpy(my_feats).cuda()
    best = 1e9
    for _ in range(a.reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        aset = ssw.align_text_batch(m, lex, d_feats, off, my_texts)
        res = [aset.utterance(k) for k in range(len(mine))]
        aset.free()
        full = gather_text_alignments(res, mine, world, rank,
                                      device=torch.device("cuda", local_rank) if a.backend == "nccl" else None)
        best = min(best, time.perf_counter() - t0)
    t = torch.tensor([best], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        import zlib
        ok = [x for x in full if x is not None]
        crc = zlib.crc32(np.concatenate([np.concatenate([x["wid"], x["state_al"].reshape(-1)]) for x in ok]).tobytes())
        tiles = all(int(x["word_al"][:, 1].sum()) == a.frames for x in ok)
        print(json.dumps({
            "workload": f"{a.utts} utterances x {a.frames} frames, texts of {a.words} words, en-us, "
                        f"features + text -> alignments, sharded over {world} GPU(s), one gather",
            "n_gpus": world, "wall_s": float(t.item()), "aligned": len(ok), "n_utts": a.utts,
            "rtf": float(t.item()) / (a.utts * a.frames / 100.0),
            "alignments_tile_their_utterances": bool(tiles), "alignment_crc32": crc}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
