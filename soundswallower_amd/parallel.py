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


class RcclComm:
    """ssw_comm_t: the library's own RCCL communicator (include/ssw_amd.h, "Multi-GPU"), for the
    one gather of the path.  Rank 0 draws the unique id (ssw_comm_unique_id) and the 128 bytes
    travel over the process group the host already has (torch.distributed here, MPI or a file
    for a C host); every rank then joins with ssw_comm_init."""

    def __init__(self, dist, world_size: int, rank: int, device_index: int):
        import ctypes as C

        import torch

        from . import _lib
        self._L = _lib.lib()
        self._c = None
        buf = C.create_string_buffer(128)
        # Rank 0 draws the id.  A failure there must not leave the other ranks waiting in the
        # broadcast: the status byte travels WITH the id and every rank raises after
        # the collective.
        status, err = 0, ""
        if rank == 0 and self._L.ssw_comm_unique_id(buf) < 0:
            status, err = 1, _lib.last_error()
        t = torch.frombuffer(bytearray(bytes([status]) + buf.raw), dtype=torch.uint8).clone()
        if world_size > 1:
            backend = dist.get_backend()
            t = t.to(torch.device("cuda", device_index)) if backend == "nccl" else t
            dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[0] != 0:
            raise RuntimeError("ssw_comm_unique_id failed on rank 0" + (": " + err if err else ""))
        idb = raw[1:]
        self._c = self._L.ssw_comm_init(idb, world_size, rank, device_index)
        if not self._c:
            raise RuntimeError("ssw_comm_init: " + _lib.last_error())
        self.world_size, self.rank = world_size, rank

    def gather(self, local: np.ndarray, counts) -> np.ndarray:
        """local int32 [n_local][3]; counts: entries of every rank.  Returns [sum(counts)][3]."""
        from . import _lib
        local = np.ascontiguousarray(local, np.int32).reshape(-1, 3)
        counts = np.ascontiguousarray(counts, np.int32)
        out = np.zeros((int(counts.sum()), 3), np.int32)
        rv = self._L.ssw_gather_alignments(self._c, local.ctypes.data, len(local),
                                           counts.ctypes.data, out.ctypes.data, None)
        if rv < 0:
            raise RuntimeError("ssw_gather_alignments: " + _lib.last_error())
        return out

    def count(self) -> int:
        """ssw_comm_count: the ranks RCCL itself says the communicator spans (ncclCommCount)."""
        return int(self._L.ssw_comm_count(self._c))

    def close(self):
        if getattr(self, "_c", None):
            self._L.ssw_comm_free(self._c)
            self._c = None

    __del__ = close


class TransportComm(RcclComm):
    """ssw_comm_from_transport: the same C gather (padding, exchange, unpacking) over an
    all-gather the HOST supplies -- what an MPI host passes (MPI_Allgather), and the seam the CPU
    tests drive ssw_gather_alignments through with several ranks.  `all_gather(send, recv,
    n_int32_per_rank)` gets two int32 numpy views (recv [world_size][n_int32_per_rank])."""

    def __init__(self, all_gather, world_size: int, rank: int):
        import ctypes as C

        from . import _lib
        self._L = _lib.lib()
        self.world_size, self.rank = world_size, rank

        def thunk(_ctx, send, recv, n):
            try:
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_int32)), shape=(int(n),))
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_int32)),
                                          shape=(world_size, int(n)))
                all_gather(s, r, int(n))
                return 0
            except Exception:      # noqa: BLE001 -- reported through the C error path
                return 1

        self._fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)(thunk)
        self._c = self._L.ssw_comm_from_transport(C.cast(self._fn, C.c_void_p), None, world_size, rank)
        if not self._c:
            raise RuntimeError("ssw_comm_from_transport: " + _lib.last_error())


def gather_alignments(local_states, n_states_per_utt, world_size: int, rank: int, *,
                      n_frames_per_utt=None, plan=None, device=None, comm=None):
    """All-gather the per-utterance state alignments.

    local_states: list of int32 arrays [n_states_u][3] for this rank's utterances, in the order
    of its shard; n_states_per_utt: state count of EVERY utterance (known to all ranks, it is a
    function of the transcripts); the shard plan is either given (`plan`, world_size lists of
    utterance ids) or recomputed from `n_frames_per_utt` -- shard_utterances is a pure function
    of the lengths, so no exchange is needed to agree on it.  Returns a list indexed by global
    utterance id.  One padded all_gather: ranks have different totals, so each sends max_total
    rows.  With `comm` (an RcclComm) the exchange is the library's C entry point
    ssw_gather_alignments -- what a C host calls -- instead of torch.distributed."""
    if plan is None:
        if n_frames_per_utt is None:
            raise ValueError("gather_alignments needs the shard plan or the utterance lengths")
        plan = shard_utterances(n_frames_per_utt, world_size)
    if len(local_states) != len(plan[rank]):
        raise ValueError(f"rank {rank} holds {len(local_states)} alignments, its shard has "
                         f"{len(plan[rank])} utterances")
    totals = [sum(int(n_states_per_utt[u]) for u in utts) for utts in plan]
    if comm is not None:
        cat = (np.concatenate([np.asarray(s, np.int32).reshape(-1, 3) for s in local_states])
               if local_states else np.zeros((0, 3), np.int32))
        if cat.shape[0] != totals[rank]:
            raise ValueError(f"rank {rank}: {cat.shape[0]} state rows, the plan says {totals[rank]}")
        rows = comm.gather(cat, totals)
        out, pos = {}, 0
        for r in range(world_size):
            for u in plan[r]:
                n = int(n_states_per_utt[u])
                out[u] = rows[pos:pos + n].copy()
                pos += n
        return [out[u] for u in sorted(out)]
    import torch
    import torch.distributed as dist

    max_total = max(totals) if totals else 0
    flat = np.zeros((max_total, 3), np.int32)
    if local_states:
        cat = np.concatenate([np.asarray(s, np.int32).reshape(-1, 3) for s in local_states])
        if cat.shape[0] != totals[rank]:
            raise ValueError(f"rank {rank}: {cat.shape[0]} state rows, the plan says {totals[rank]}")
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
    return [out[u] for u in sorted(out)]


def gather_text_alignments(local, local_utts, world_size: int, rank: int, device=None):
    """All-gather alignments made from TEXT (forced_align_batch): unlike gather_alignments the
    entry counts are not known in advance -- the first pass decides which fillers and alternates
    each utterance contains -- so the ranks exchange their buffer lengths first, then one padded
    all_gather of the flattened results.

    local: one item per utterance of local_utts: None (not aligned) or a dict with int32 arrays
    wid [n_w], word_al [n_w][3], cipid [n_p], parent [n_p], phone_al [n_p][3], state_al
    [3 n_p][3] (AlignmentSet.utterance).  Returns a list indexed by global utterance id."""
    import torch
    import torch.distributed as dist

    parts = []
    for u, a in zip(local_utts, local):
        if a is None:
            parts.append(np.array([int(u), -1, 0], np.int32))
            continue
        n_w, n_p = len(a["wid"]), len(a["cipid"])
        parts.append(np.concatenate([
            np.array([int(u), n_w, n_p], np.int32), np.asarray(a["wid"], np.int32),
            np.asarray(a["word_al"], np.int32).reshape(-1), np.asarray(a["cipid"], np.int32),
            np.asarray(a["parent"], np.int32), np.asarray(a["phone_al"], np.int32).reshape(-1),
            np.asarray(a["state_al"], np.int32).reshape(-1)]))
    flat = np.concatenate(parts) if parts else np.zeros(0, np.int32)
    n = torch.tensor([len(flat)], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    lens = [torch.zeros_like(n) for _ in range(world_size)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    buf = np.zeros(max(max(lens), 1), np.int32)
    buf[:len(flat)] = flat
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(world_size)]
    dist.all_gather(bufs, t)
    out = {}
    for r in range(world_size):
        v = bufs[r].cpu().numpy()[:lens[r]]
        pos = 0
        while pos < len(v):
            u, n_w, n_p = int(v[pos]), int(v[pos + 1]), int(v[pos + 2])
            pos += 3
            if n_w < 0:
                out[u] = None
                continue
            a = {}
            for key, cnt, cols in (("wid", n_w, 0), ("word_al", n_w, 3), ("cipid", n_p, 0),
                                   ("parent", n_p, 0), ("phone_al", n_p, 3), ("state_al", 3 * n_p, 3)):
                k = cnt * (cols or 1)
                a[key] = v[pos:pos + k].reshape(cnt, cols).copy() if cols else v[pos:pos + k].copy()
                pos += k
            out[u] = a
    return [out[u] for u in sorted(out)]
