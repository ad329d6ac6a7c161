#!/usr/bin/env python3
"""bench.py -- senone-frames/s of the PTM scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (ssw_score_batch: density/top-N kernel + senone kernel)
over one batch of synthetic features already resident in HBM: BASELINE.json configs[1],
4096 frames of 39-dim features as 16 utterances x 256 frames, en-us PTM model.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank scores its own shard of
utterances -- the path partitions by utterance, there is no data-path collective -- and the
reported value is the whole-job rate over the max-over-ranks time (weak scaling).

Prints ONE JSON line on rank 0, including `roofline` (dominant kernel, HIP-event timed on the
launch stream) and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same
workload on the host cores of this box).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_UTTS = 16
UTT_FRAMES = 256
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md
N_SIMD = 1024                   # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
PMC_FILE = "r01_l_pmc.json"    # committed rocprofv3 --pmc measurement of this exact step


def algorithmic_bytes(n_sen, n_feat, topn, n_cb, n_density, veclen_total, batch):
    """SURVEY.md 8(d) touched-bytes accounting, per frame, and its split over the two kernels."""
    feat_b = 4 * veclen_total
    out_b = 2 * n_sen
    mixw_b = n_sen * n_feat * topn
    gauss_block = n_cb * n_density * (2 * veclen_total + n_feat) * 4
    topn_b = n_cb * n_feat * (topn + 4 * topn)          # packed cw + int32 scores between kernels
    whole = feat_b + out_b + mixw_b + gauss_block / batch
    return {"path": whole,
            "topn_kernel": feat_b + gauss_block / batch + topn_b,
            "senone_kernel": topn_b + mixw_b + out_b}


def align_config3(ssw, model, means, torch, n_utts=256, n_frames=1000, n_phones=150, reps=3):
    """BASELINE configs[2]: score and force-align 256 synthetic utterances x 1000 frames x 150
    phones (the "align RTF" part of the metric); best of `reps` passes."""
    from soundswallower_amd.synth import synth_alignment_task
    sseq = model.table("sseq").reshape(-1, 3)
    pssid, ptmat = model.table("phone_ssid"), model.table("phone_tmat")
    feats = np.concatenate([ssw.synth_features(means, n_frames, 12345 + u) for u in range(n_utts)])
    senid, tmat = [], []
    for u in range(n_utts):
        s_, t_, _ = synth_alignment_task(sseq, pssid, ptmat, model.n_ciphone, n_phones, 777 + u)
        senid.append(s_)
        tmat.append(t_)
    senid, tmat = np.concatenate(senid), np.concatenate(tmat)
    frame_off = (np.arange(n_utts + 1) * n_frames).astype(np.int32)
    phone_off = (np.arange(n_utts + 1) * n_phones).astype(np.int32)
    total = n_utts * n_frames
    d_feats = torch.from_numpy(feats).cuda()
    d_scr = torch.empty((total, model.n_sen), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.score_batch_device(d_feats, total, frame_off, d_scr, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st, status = model.align_batch(d_scr.data_ptr(), frame_off, phone_off, senid, tmat)
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    tiles = all(st[phone_off[u] * 3:phone_off[u + 1] * 3, 1].sum() == n_frames
                for u in range(n_utts) if status[u] == 0)
    return {"workload": f"{n_utts} utterances x {n_frames} frames x {n_phones} phones, en-us: PTM "
                        f"scoring + forced alignment (BASELINE configs[2])",
            "score_ms": best[1] * 1e3, "align_ms": best[2] * 1e3,
            "rtf": best[0] / (total / 100.0), "utt_frames_per_s": total / best[0],
            "aligned": int((status == 0).sum()), "alignments_tile_their_utterances": bool(tiles)}


def cpu_baseline(model_dir, feats, utt_off, target_s=12.0):
    """Time the CPU oracle (scalar C restatement of the reference path) on a bounded sample."""
    from oracle import oracle as O
    m = O.Model(model_dir)
    n_done, n_utt, t0 = 0, 0, time.perf_counter()
    n_all = len(utt_off) - 1
    while time.perf_counter() - t0 < target_s:  # cycle over the batch until ~target_s of CPU work
        u = n_utt % n_all
        m.ptm_score_utt(feats[utt_off[u]:utt_off[u + 1]])
        n_done += int(utt_off[u + 1] - utt_off[u])
        n_utt += 1
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_utt} utterance passes x {UTT_FRAMES} frames cycling over the bench batch's "
                      f"{n_all} utterances ({n_done} frames, {dt:.1f} s, 1 thread of the CPU "
                      f"oracle oracle/ssw_oracle.c, a scalar C port of the reference path)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="en-us")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-align", action="store_true",
                    help="skip the config-3 alignment pass that fills the `align` object")
    ap.add_argument("--utts", type=int, default=N_UTTS,
                    help="utterances (x256 frames) per GPU per step; the default is BASELINE.json "
                         "configs[1], 4096 frames")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scoring path has no CPU fallback")
    # SSW_BENCH_BACKEND=gloo + SSW_BENCH_DEVICE=0 let several ranks share one GPU, to exercise the
    # multi-rank control flow on a 1-GPU box; the driver's runs use nccl (= RCCL), one GPU per rank
    backend = os.environ.get("SSW_BENCH_BACKEND", "nccl")
    if "SSW_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SSW_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import soundswallower_amd as ssw
    from soundswallower_amd import _lib
    from soundswallower_amd.parallel import shard_utterances
    from soundswallower_amd.synth import read_raw_means as raw_means

    if rank == 0:          # one rank compiles (a no-op when the .so is current); the rest wait
        _lib.build()
    if dist:
        dist.barrier()
    mdir = ssw.model_dir(args.model)
    model = ssw.Model(mdir, config={"device": local_rank})
    means = raw_means(mdir)

    # this rank's shard of the global utterance list (weak scaling: N_UTTS per rank)
    global_utts = list(range(args.utts * world))
    mine = shard_utterances([UTT_FRAMES] * len(global_utts), world)[rank]
    feats = np.concatenate([ssw.synth_features(means, UTT_FRAMES, 12345 + u) for u in mine])
    utt_off = (np.arange(len(mine) + 1) * UTT_FRAMES).astype(np.int32)
    n_frames = feats.shape[0]

    d_feats = torch.from_numpy(feats).cuda()
    d_out = torch.empty((n_frames, model.n_sen), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        model.score_batch_device(d_feats, n_frames, utt_off, d_out, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations from HIP events recorded on the same stream, same steps again
    model.set_kernel_timing(True)
    k_ms = np.zeros(2)
    n_prof = max(1, min(args.steps, 50))
    for _ in range(n_prof):
        step()
        k_ms += np.array(model.kernel_timing())
    model.set_kernel_timing(False)
    k_ms /= n_prof

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    ab = algorithmic_bytes(model.n_sen, model.n_feat, model.topn, model.n_cb, model.n_density,
                           model.veclen_total, n_frames)
    names = ("topn_kernel", "senone_kernel")
    total_frames = n_frames * world * args.steps
    fps = total_frames / elapsed
    flagged, pairs = model.last_stats()
    # roofline of the hot path = the launches of one step (top-N pass incl. its exact fix-up,
    # senone pass); algorithmic bytes = SURVEY 8(d)'s 72,345 B/frame x frames per launch set
    path_ms = float(k_ms.sum())
    achieved = ab["path"] * n_frames / (path_ms * 1e-3) / 1e9
    traffic = valu_instr = None
    tfile = os.path.join(ROOT, "profiles", PMC_FILE)
    if os.path.exists(tfile) and n_frames == 4096 and args.model == "en-us":
        with open(tfile) as fh:      # PMC counters cannot be read from inside this process;
            pmc = json.load(fh)      # committed rocprofv3 measurement of this step
        traffic = pmc["hbm_bytes"]
        valu_instr = pmc["valu_wave_instr_per_step"]
    per_kernel = {
        nm: {"ms": float(k_ms[i]), "algorithmic_bytes_per_frame": ab[nm],
             "achieved_GBps": ab[nm] * n_frames / (k_ms[i] * 1e-3) / 1e9,
             "hbm_frac": ab[nm] * n_frames / (k_ms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        for i, nm in enumerate(names)}
    out = {
        "metric": "senone-frames/sec (en-us PTM)",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32+i32",
        "data": "synthetic",
        "config": {"workload": f"PTM senone scoring, {args.model}, {args.utts} utterances x "
                               f"{UTT_FRAMES} frames = {n_frames} frames per GPU per step, "
                               f"39-dim features resident in HBM, compallsen=yes, topn=4",
                   "senones": model.n_sen, "codebooks": model.n_cb,
                   "parallelism": f"utt-shard x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "PTM path of one step: ptm_topn_frames + ptm_topn_fixup + ptm_senone",
                     "kernel_ms": path_ms, "algorithmic_bytes_per_frame": ab["path"],
                     "note": "VALU-issue bound, not HBM bound (DESIGN.md section 5): see valu_frac"},
        "kernels": per_kernel,
        # share of the VALU issue slots the step uses if every wave64 instruction took the
        # nominal 4 cycles (SQ_INSTS_VALU from the PMC file); add/sub/mul/fma/logic issue in 2 on
        # gfx950 (tools/microbench/valu_rate.hip), so this over-counts the senone kernel
        "valu_frac": (valu_instr * 4.0 / N_SIMD / CLOCK_HZ / (path_ms * 1e-3)
                      if valu_instr else None),
        "exact_pass_share": flagged / max(pairs, 1),
    }
    if not args.no_align and world == 1 and args.model == "en-us":
        out["align"] = align_config3(ssw, model, means, torch)
        # the same job from TEXT: first pass (which fillers / alternates, word frames) + the
        # rest of decoder_alignment, tools/bench_first_pass.py
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import bench_first_pass
        lex = ssw.Lexicon(model, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
        out["text_align"] = bench_first_pass.run(ssw, model, lex, torch)
    if not args.no_cpu_baseline and world == 1:   # timed on rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(mdir, feats, utt_off)
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
