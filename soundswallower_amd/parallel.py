"""Utterance sharding across the GPUs of one node.

Utterances are independent (every utterance starts from the reset top-N history), so the path
shards with no data-path collective: sort by frame count, deal round-robin to ranks, replicate
the 3.7 MB model on every GPU.  The only exchange is one gather of the final alignment arrays
(a few KB per utterance) over RCCL (`torch.distributed` backend "nccl" on ROCm; "gloo" in the
CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_utterances(n_frames_per_utt, world_size: int):
    """Longest-first round-robin deal.  Returns world_size lists of utterance indices; the
    assignment is a pure function of the lengths, so every rank computes the same plan."""
    order = sorted(range(len(n_frames_per_utt)), key=lambda u: (-int(n_frames_per_utt[u]), u))
    shards = [[] for _ in range(world_size)]
    for k, u in enumerate(order):
        shards[k % world_size].append(u)
    return shards


def gather_alignments(local_states, local_utts, n_states_per_utt, world_size: int, rank: int,
                      device=None):
    """All-gather the per-utterance state alignments.

    local_states: list of int32 arrays [n_states_u][3] for this rank's utterances (in the order
    of local_utts); n_states_per_utt: state count of EVERY utterance (known to all ranks, it is
    a function of the transcripts).  Returns a list indexed by global utterance id.
    One padded all_gather: ranks have different totals, so each sends max_total rows."""
    import torch
    import torch.distributed as dist

    shards_sizes = [0] * world_size
    plan = shard_utterances_by_list(local_utts, world_size, rank)
    totals = [sum(int(n_states_per_utt[u]) for u in utts) for utts in plan]
    max_total = max(totals) if totals else 0
    flat = np.zeros((max_total, 3), np.int32)
    if local_states:
        cat = np.concatenate([np.asarray(s, np.int32).reshape(-1, 3) for s in local_states])
        flat[:cat.shape[0]] = cat
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(world_size)]
    dist.all_gather(bufs, t)
    out = {}
    for r in range(world_size):
        rows = bufs[r].cpu().numpy()
        pos = 0
        for u in plan[r]:
            n = int(n_states_per_utt[u])
            out[u] = rows[pos:pos + n].copy()
            pos += n
    del shards_sizes
    return [out[u] for u in sorted(out)]


def shard_utterances_by_list(local_utts, world_size, rank):
    """Exchange every rank's utterance list (tiny) so all ranks know the full plan."""
    import torch.distributed as dist

    plan = [None] * world_size
    dist.all_gather_object(plan, [int(u) for u in local_utts])
    return plan
