#!/usr/bin/env python3
"""BASELINE config 3 / 5 timing: N synthetic utterances x F frames, ~P phones each:
PTM scoring of every frame, then forced-alignment Viterbi of every utterance; one MI355X, or
under `python -m torch.distributed.run --nproc-per-node N` the utterances dealt over N GPUs with
one RCCL gather of the final alignments (config 5).  The job itself is
soundswallower_amd/jobs.py (bench.py runs the same code for its `config5` object).
Prints one JSON line (utterance-frames/s, align RTF = wall / audio seconds at 100 frames/s)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundswallower_amd as ssw  # noqa: E402
from soundswallower_amd import _lib, jobs  # noqa: E402
from soundswallower_amd.synth import read_raw_means  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256, help="utterances in the whole job")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--phones", type=int, default=150)
    ap.add_argument("--chunk", type=int, default=jobs.CHUNK_UTTS,
                    help="utterances scored and aligned per call")
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
    dist = device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)
    if rank == 0:
        _lib.build()
    if dist:
        dist.barrier()
    mdir = ssw.model_dir("en-us")
    m = ssw.Model(mdir, config={"device": local_rank})
    out = jobs.run_config5(m, read_raw_means(mdir), dist, rank, world, device, reps=a.reps,
                           n_utts=a.utts, n_frames=a.frames, n_phones=a.phones,
                           chunk_utts=a.chunk)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
