#!/usr/bin/env python3
"""BASELINE config 3 / 5 timing: N synthetic utterances x F frames, ~P phones each:
PTM scoring of every frame, then forced-alignment Viterbi of every utterance; one MI355X, or
under `python -m torch.distributed.run --nproc-per-node N` the utterances dealt over N GPUs with
one RCCL gather of the final alignments (config 5).
Prints one JSON line (utterance-frames/s, align RTF = wall / audio seconds at 100 frames/s)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_alignment_task, synth_features  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256, help="utterances in the whole job")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--phones", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for the gather (nccl = RCCL; gloo lets two "
                         "ranks share one GPU on a 1-GPU box to exercise the multi-rank flow)")
    ap.add_argument("--device", type=int, default=None, help="GPU to use (default LOCAL_RANK)")
    a = ap.parse_args()
    # config 5: launched as `python -m torch.distributed.run --nproc-per-node N ... tools/bench_align.py`;
    # the job's utterances are dealt to the ranks (strong scaling), every rank scores and aligns its
    # own, and the final state alignments are gathered once over RCCL
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.device is not None:
        local_rank = a.device
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    from soundswallower_amd.parallel import gather_alignments, shard_utterances
    if rank == 0:
        _lib.build()
    if dist:
        dist.barrier()
    L = _lib.lib()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir, config={"device": local_rank})
    means = read_raw_means(mdir)
    sseq = m.table("sseq").reshape(-1, 3)
    pssid, ptmat = m.table("phone_ssid"), m.table("phone_tmat")
    mine = shard_utterances([a.frames] * a.utts, world)[rank]
    n_mine = len(mine)
    n_total = n_mine * a.frames
    feats = np.concatenate([synth_features(means, a.frames, 12345 + u) for u in mine])
    senid, tmat = [], []
    for u in mine:
        s, t, _ = synth_alignment_task(sseq, pssid, ptmat, m.n_ciphone, a.phones, 777 + u)
        senid.append(s)
        tmat.append(t)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    frame_off = (np.arange(n_mine + 1) * a.frames).astype(np.int32)
    phone_off = (np.arange(n_mine + 1) * a.phones).astype(np.int32)
    d_feats = m.to_device(feats)
    d_scr = L.ssw_device_malloc(C.c_size_t(n_total * m.n_sen * 2))
    best = {"score_s": 1e9, "align_s": 1e9, "gather_s": 0.0, "wall_s": 1e9}
    gathered = None
    for _ in range(a.reps):
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        m.score_batch_device(d_feats, n_total, frame_off, d_scr)
        L.ssw_device_synchronize()
        t1 = time.perf_counter()
        st, status = m.align_batch(d_scr, frame_off, phone_off, senid, tmat)
        t2 = time.perf_counter()
        t3 = t2
        if dist:
            import torch
            per_utt = [st[phone_off[k] * 3:phone_off[k + 1] * 3] for k in range(n_mine)]
            gathered = gather_alignments(
                per_utt, mine, [a.phones * 3] * a.utts, world, rank,
                device=torch.device("cuda", local_rank) if a.backend == "nccl" else None)
            t3 = time.perf_counter()
        if t3 - t0 < best["wall_s"]:
            best = {"score_s": t1 - t0, "align_s": t2 - t1, "gather_s": t3 - t2, "wall_s": t3 - t0}
    ok = int((status == 0).sum())
    tiles_ok = bool(all((st[phone_off[u] * 3:phone_off[u + 1] * 3, 1].sum() == a.frames)
                        for u in range(n_mine) if status[u] == 0))
    wall = best["wall_s"]
    if dist:
        import torch
        t = torch.tensor([wall, float(ok), float(tiles_ok)], dtype=torch.float64,
                         device="cuda" if a.backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        wall, ok = float(tmax[0]), int(t[1])
        tiles_ok = bool(t[2] == world) and len(gathered) == a.utts
    if rank == 0:
        import zlib
        # same value whatever the number of ranks: utterances in global order
        per_global = gathered if dist else [st[phone_off[k] * 3:phone_off[k + 1] * 3]
                                            for k in np.argsort(mine)]
        crc = zlib.crc32(np.ascontiguousarray(np.concatenate(per_global), np.int32).tobytes())
        job_frames = a.utts * a.frames
        print(json.dumps({
            "workload": f"{a.utts} utterances x {a.frames} frames x {a.phones} phones, en-us, "
                        f"sharded over {world} GPU(s)",
            "n_gpus": world,
            "score_s": best["score_s"], "align_s": best["align_s"], "gather_s": best["gather_s"],
            "wall_s": wall,
            "score_frames_per_s": n_total / best["score_s"],
            "align_utt_frames_per_s": n_total / best["align_s"],
            "job_utt_frames_per_s": job_frames / wall,
            "align_rtf": wall / (job_frames / 100.0),
            "aligned_ok": ok, "n_utts": a.utts,
            "tiles_ok": tiles_ok, "alignment_crc32": crc,
        }))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
