"""The sharded batch-alignment job of BASELINE.json configs[4] ("config 5" in SURVEY.md 8(d)):
N synthetic utterances x F frames x P phones, dealt over the ranks of one node; every rank
scores its utterances (ssw_score_batch) and force-aligns them (ssw_align_batch) -- the whole shard
at once while its scores fit in HBM, in chunks beyond that -- and
the final state alignments are gathered once (parallel.gather_alignments: RCCL on a GPU node,
gloo in the CPU tests).  Used by bench.py (`config5` object of the bench line),
tools/bench_align.py and tests/test_gpu_config5.py, so all three run the same code.

Inputs follow SURVEY 8(d): features of utterance u from the LCG seed 12345 + u, its phone
string from seed 777 + u -- the same seeds as tests/golden (config 3), so the first utterances
of the job have committed checksums.
"""
from __future__ import annotations

import ctypes
import os
import time
import zlib

import numpy as np

from .parallel import gather_alignments, shard_utterances
from .synth import synth_alignment_task, synth_features

N_UTTS, N_FRAMES, N_PHONES, CHUNK_UTTS = 2048, 1000, 150, 256
RESIDENT_BYTES = 64 << 30   # a shard whose senone scores take less stays resident as a whole ...
RESIDENT_SHARE = 0.5        # ... if that is at most this share of the device's FREE memory (the
                            # scoring and alignment workspaces need about half as much again)


class Config5Shard:
    """One rank's share of the job with its inputs resident in HBM (built outside any timed
    region): features of its utterances on the device, phone strings on the host."""

    def __init__(self, model, means, rank=0, world=1, n_utts=N_UTTS, n_frames=N_FRAMES,
                 n_phones=N_PHONES, chunk_utts=CHUNK_UTTS):
        self.model, self.rank, self.world = model, rank, world
        self.n_utts, self.n_frames, self.n_phones = n_utts, n_frames, n_phones
        self.mine = shard_utterances([n_frames] * n_utts, world)[rank]
        # Scoring runs a chunk ahead of alignment (score_and_align).  The alignment kernel is
        # latency-bound -- one dependent step per frame: 2.4 ms for 64 utterances x 1000 frames,
        # 2.2 ms for 256 -- so cutting a shard finer than needed ADDS alignment time: a 256-
        # utterance shard (one of 8 ranks) takes 7.7 ms as one chunk and 11.7 ms as four
        # (measured).  Chunks of `chunk_utts`, the last one ragged.
        self.chunk_utts = max(1, min(chunk_utts, max(1, len(self.mine))))
        sseq = model.table("sseq").reshape(-1, model.n_emit_state)
        pssid, ptmat = model.table("phone_ssid"), model.table("phone_tmat")
        senid, tmat = [], []
        for u in self.mine:
            s, t, _ = synth_alignment_task(sseq, pssid, ptmat, model.n_ciphone, n_phones, 777 + u)
            senid.append(s)
            tmat.append(t)
        self.senid = (np.concatenate(senid) if senid
                      else np.zeros((0, model.n_emit_state), np.uint16))
        self.tmat = np.concatenate(tmat) if tmat else np.zeros(0, np.int16)
        self.d_feats, self.d_scr = None, None
        self.resident = False
        self.fallback_note = None
        # Round 5: compact score rows (include/ssw_amd.h).  The phone strings are known before
        # the shard is scored, so the scorer stores each utterance's own 450 states' scores only
        # (0.9 KB instead of 10 KB per frame: 1.8 GB instead of 21 GB for the whole job) and the
        # alignment kernel reads them coalesced.  SSW_JOB_ROWS=full keeps rounds 3-4's full rows.
        self.compact = os.environ.get("SSW_JOB_ROWS", "compact") != "full"
        self.d_compact = None
        self.s_score = self.s_align = None
        if self.mine:
            # uploaded chunk by chunk: the host never holds more than one chunk of features
            row = model.veclen_total * 4
            self.d_feats = model.device_malloc(len(self.mine) * n_frames * row)
            for k, u in enumerate(self.mine):
                f = synth_features(means, n_frames, 12345 + u)
                model._L.ssw_memcpy_h2d(self.d_feats + k * n_frames * row,
                                        f.ctypes.data, f.nbytes)
            # Round 3: when the shard's scores fit in HBM (21 GB for 2048 x 1000 frames; the GPU
            # has 288) the whole shard is scored, then aligned in ONE call: the alignment kernel
            # is latency-bound -- 1.3 ms per 1000 frames for 256 utterances, and barely more for
            # 2048 (six waves per SIMD instead of one) -- so eight calls beside the scoring of the
            # next chunk (each three times slower for sharing the chip, and slowing the scoring)
            # cost more than one call at the end.  Beyond RESIDENT_BYTES: chunks, ping-pong.
            if self.compact:
                self.frame_off = (np.arange(len(self.mine) + 1) * n_frames).astype(np.int32)
                self.phone_off = (np.arange(len(self.mine) + 1) * n_phones).astype(np.int32)
                self.d_compact = model.device_malloc(
                    len(self.mine) * n_frames * ((3 * n_phones + 1) & ~1) * 2)
                self.resident = True
                self.s_score = model._L.ssw_stream_create()
                self.s_align = model._L.ssw_stream_create()
                return
            total = len(self.mine) * n_frames * model.n_sen * 2
            free_b = ctypes.c_size_t(0)
            if model._L.ssw_device_mem_info(ctypes.byref(free_b), None) != 0:
                free_b.value = 0            # (unknown: take the chunked path)
            self.resident = (total <= RESIDENT_BYTES and total <= RESIDENT_SHARE * free_b.value
                             and chunk_utts >= CHUNK_UTTS)
            nbytes = self.chunk_utts * n_frames * model.n_sen * 2
            if self.resident:               # a smaller or shared GPU: fall back to the chunks
                from .api import SswError
                try:
                    self.d_scr = model.device_malloc(total)
                except SswError as e:       # hipMalloc said no (ssw_device_malloc clears the error)
                    self.resident = False
                    self.fallback_note = f"resident allocation of {total} bytes failed ({e}): chunks"
            if not self.resident:
                self.d_scr = model.device_malloc(2 * nbytes)
            self.d_scr2 = (self.d_scr, self.d_scr + (0 if self.resident else nbytes))
            self.s_score = model._L.ssw_stream_create()
            self.s_align = model._L.ssw_stream_create()

    def close(self):
        for p in (self.d_feats, self.d_scr, self.d_compact):
            if p:
                self.model.device_free(p)
        self.d_feats = self.d_scr = self.d_compact = None
        for st in (self.s_score, self.s_align):
            if st:
                self.model._L.ssw_stream_destroy(st)
        self.s_score = self.s_align = None

    def score_and_align(self):
        """Scores and aligns the shard chunk by chunk.  Scoring is asynchronous on a stream of
        its own and one chunk ahead of the (synchronous, latency-bound) alignment, which leaves
        most of the chip to it: chunk k + 1 is scored while chunk k is aligned, into the other
        of two score buffers.  Returns (states int32 [n_mine * 3P][3], status int32 [n_mine],
        score_s, align_s) -- score_s is what the host still waits for scores, align_s the
        alignment calls."""
        m, F, P, L = self.model, self.n_frames, self.n_phones, self.model._L
        n_mine = len(self.mine)
        states = np.zeros((n_mine * P * 3, 3), np.int32)
        status = np.zeros(n_mine, np.int32)
        t_score = t_align = 0.0
        row = m.veclen_total * 4
        chunks = [(c0, min(n_mine, c0 + self.chunk_utts)) for c0 in range(0, n_mine, self.chunk_utts)]
        if not chunks:
            return states, status, 0.0, 0.0
        if self.compact:
            # the plan (which states each utterance reads, where their scores go) is part of the
            # job: built inside the timed region, counted with the scoring
            ta = time.perf_counter()
            plan = m.compact_plan(self.frame_off, self.phone_off, self.senid, stream=self.s_score)
            try:
                assert plan.nbytes == n_mine * F * ((3 * P + 1) & ~1) * 2
                m.score_batch_compact(self.d_feats, plan, self.d_compact, self.s_score)
                L.ssw_stream_synchronize(self.s_score)
                tb = time.perf_counter()
                st, stat = m.align_batch_compact(plan, self.d_compact, self.tmat,
                                                 stream=self.s_align)
                tc = time.perf_counter()
            finally:
                plan.free()
            return st, stat, tb - ta, tc - tb
        if self.resident:
            frame_off = (np.arange(n_mine + 1) * F).astype(np.int32)
            phone_off = (np.arange(n_mine + 1) * P).astype(np.int32)
            ta = time.perf_counter()
            m.score_batch_device(self.d_feats, n_mine * F, frame_off, self.d_scr, self.s_score)
            L.ssw_stream_synchronize(self.s_score)
            tb = time.perf_counter()
            st, stat = m.align_batch(self.d_scr, frame_off, phone_off, self.senid, self.tmat,
                                     stream=self.s_align)
            tc = time.perf_counter()
            return st, stat, tb - ta, tc - tb

        def launch_score(k):
            c0, c1 = chunks[k]
            n = c1 - c0
            frame_off = (np.arange(n + 1) * F).astype(np.int32)
            m.score_batch_device(self.d_feats + c0 * F * row, n * F, frame_off, self.d_scr2[k & 1],
                                 self.s_score, share_device=True)

        launch_score(0)
        for k, (c0, c1) in enumerate(chunks):
            n = c1 - c0
            frame_off = (np.arange(n + 1) * F).astype(np.int32)
            phone_off = (np.arange(n + 1) * P).astype(np.int32)
            ta = time.perf_counter()
            L.ssw_stream_synchronize(self.s_score)          # chunk k's scores are complete
            tb = time.perf_counter()
            if k + 1 < len(chunks):
                launch_score(k + 1)                           # runs beside the alignment below
            st, stat = m.align_batch(self.d_scr2[k & 1], frame_off, phone_off,
                                     self.senid[c0 * P:c1 * P], self.tmat[c0 * P:c1 * P],
                                     stream=self.s_align)
            tc = time.perf_counter()
            states[c0 * P * 3:c1 * P * 3] = st
            status[c0:c1] = stat
            t_score += tb - ta
            t_align += tc - tb
        return states, status, t_score, t_align

    def run(self, dist=None, device=None, comm=None):
        """One pass of the whole job on this rank: score + align its shard, then the single
        gather.  Returns a dict; `per_utt` (list indexed by global utterance id) on every rank."""
        P = self.n_phones
        t0 = time.perf_counter()
        states, status, t_score, t_align = self.score_and_align()
        t1 = time.perf_counter()
        local = [states[k * P * 3:(k + 1) * P * 3] for k in range(len(self.mine))]
        if dist is not None and self.world > 1:
            per_utt = gather_alignments(local, [P * 3] * self.n_utts, self.world, self.rank,
                                        n_frames_per_utt=[self.n_frames] * self.n_utts,
                                        device=device, comm=comm)
        else:
            per_utt = [None] * self.n_utts
            for k, u in enumerate(self.mine):
                per_utt[u] = local[k]
        t2 = time.perf_counter()
        tiles = all(int(local[k][:, 1].sum()) == self.n_frames
                    for k in range(len(self.mine)) if status[k] == 0)
        return {"per_utt": per_utt, "status": status, "score_s": t_score, "align_s": t_align,
                "gather_s": t2 - t1, "wall_s": t2 - t0, "tiles": bool(tiles),
                "aligned": int((status == 0).sum())}


def alignment_crc(per_utt) -> int:
    """CRC-32 of the state alignments in global utterance order: the same value whatever the
    number of ranks the job ran on."""
    c = 0
    for a in per_utt:
        c = zlib.crc32(np.ascontiguousarray(a, np.int32).tobytes(), c)
    return c & 0xFFFFFFFF


def run_config5(model, means, dist=None, rank=0, world=1, device=None, reps=2, n_utts=N_UTTS,
                n_frames=N_FRAMES, n_phones=N_PHONES, chunk_utts=CHUNK_UTTS, keep=4, comm=None):
    """The whole job, `reps` times, best wall (max over ranks) reported.  Every rank must call
    it.  Returns on every rank a dict of job-level numbers (rank 0's are the ones to print);
    `first_states_crc` = CRC-32 of each of the first `keep` utterances' state alignments,
    comparable with the committed config-3 checksums under tests/golden/."""
    shard = Config5Shard(model, means, rank, world, n_utts, n_frames, n_phones, chunk_utts)
    best = None
    try:
        for _ in range(reps):
            if dist is not None and world > 1:
                dist.barrier()
            r = shard.run(dist, device, comm)
            wall, ok, tiles = r["wall_s"], r["aligned"], int(r["tiles"])
            if dist is not None and world > 1:
                import torch
                dev = device if device is not None else "cpu"
                tmax = torch.tensor([wall, r["score_s"], r["align_s"], r["gather_s"]],
                                    dtype=torch.float64, device=dev)
                tmin = tmax.clone()         # the fastest rank's figures: the spread shows imbalance
                tsum = torch.tensor([float(ok), float(tiles)], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
                dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                wall, score_s, align_s, gather_s = (float(x) for x in tmax.cpu())
                lo = [float(x) for x in tmin.cpu()]
                ok, tiles = int(tsum[0].item()), int(tsum[1].item())
            else:
                score_s, align_s, gather_s = r["score_s"], r["align_s"], r["gather_s"]
                lo = [wall, score_s, align_s, gather_s]
            if best is None or wall < best["wall_s"]:
                best = {"wall_s": wall, "score_s": score_s, "align_s": align_s,
                        "gather_s": gather_s, "aligned": ok, "tiles": tiles == world,
                        "min": lo, "per_utt": r["per_utt"]}
    finally:
        shard.close()
    job_frames = n_utts * n_frames
    per_utt = best.pop("per_utt")
    complete = all(a is not None for a in per_utt)
    return {
        "workload": f"{n_utts} utterances x {n_frames} frames x {n_phones} phones, en-us, dealt "
                    f"over {world} rank(s), "
                    + ("every shard scored as a whole into compact rows (its utterances' own "
                       "states only) and aligned in one call"
                       if getattr(shard, "compact", False)
                       else "every shard scored as a whole and aligned in one call"
                       if getattr(shard, "resident", False)
                       else f"in chunks of {shard.chunk_utts} utterances")
                    + ": PTM scoring + forced alignment per rank, one gather of the state "
                      "alignments (BASELINE configs[4])",
        "n_ranks": world, "n_utts": n_utts,
        "wall_ms": best["wall_s"] * 1e3, "score_ms": best["score_s"] * 1e3,
        "align_ms": best["align_s"] * 1e3, "gather_ms": best["gather_s"] * 1e3,
        # max over ranks above (what the wall clock waits for); the fastest rank's, for imbalance
        "per_rank_min": {"wall_ms": best["min"][0] * 1e3, "score_ms": best["min"][1] * 1e3,
                         "align_ms": best["min"][2] * 1e3, "gather_ms": best["min"][3] * 1e3},
        "fallback": getattr(shard, "fallback_note", None),
        "job_utt_frames_per_s": job_frames / best["wall_s"],
        "align_rtf": best["wall_s"] / (job_frames / 100.0),
        "aligned": best["aligned"], "alignments_tile_their_utterances": bool(best["tiles"]),
        "gathered_all": bool(complete),
        "alignment_crc32": alignment_crc(per_utt) if complete else None,
        "first_states_crc": [alignment_crc([a]) for a in per_utt[:keep]] if complete else None,
    }
