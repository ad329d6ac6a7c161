"""N > 1 path on CPU: world_size 2 over gloo.  Utterance sharding plan + the single gather of
final alignments; the per-rank compute is replaced by the oracle here, the
collective and the plan are the product code."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from soundswallower_amd.parallel import shard_utterances


def test_shard_plan_is_balanced_and_complete():
    lens = [1000, 10, 500, 500, 990, 20, 30, 980]
    for w in (1, 2, 3, 4, 8):
        plan = shard_utterances(lens, w)
        flat = sorted(u for p in plan for u in p)
        assert flat == list(range(len(lens)))
        loads = [sum(lens[u] for u in p) for p in plan]
        assert max(loads) - min(loads) <= max(lens)
    assert shard_utterances([], 2) == [[], []]
    assert shard_utterances([5], 4) == [[0], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_states, q):
    import torch.distributed as dist
    from soundswallower_amd.parallel import gather_alignments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [7 * (u + 1) for u in range(len(n_states))]
    mine = shard_utterances(lens, world)[rank]
    local = []
    for u in mine:  # stand-in alignment: deterministic function of (utterance, state)
        n = n_states[u]
        local.append(np.stack([np.arange(n) * (u + 1), np.full(n, u + 1), -np.arange(n) - u], 1)
                     .astype(np.int32))
    full = gather_alignments(local, n_states, world, rank, n_frames_per_utt=lens)
    if rank == 0:
        q.put([a.tolist() for a in full])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_alignments_world2():
    n_states = [9, 3, 12, 6, 15]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_states, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(got) == len(n_states)
    for u, n in enumerate(n_states):
        exp = np.stack([np.arange(n) * (u + 1), np.full(n, u + 1), -np.arange(n) - u], 1)
        assert np.array_equal(np.array(got[u]), exp)


def _text_worker(rank, world, port, n_utts, q):
    import torch.distributed as dist
    from soundswallower_amd.parallel import gather_text_alignments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_utterances([100 + 7 * u for u in range(n_utts)], world)[rank]
    local = [_fake_text_alignment(u) for u in mine]
    full = gather_text_alignments(local, mine, world, rank)
    if rank == 0:
        q.put([None if a is None else {k: v.tolist() for k, v in a.items()} for a in full])
    dist.barrier()
    dist.destroy_process_group()


def _fake_text_alignment(u):
    """stand-in for AlignmentSet.utterance(u): sizes vary per utterance, one is not aligned"""
    if u == 3:
        return None
    n_w, n_p = 2 + u % 3, 4 + u
    r = np.random.default_rng(u)
    return {"wid": r.integers(0, 1000, n_w).astype(np.int32),
            "word_al": r.integers(-500, 500, (n_w, 3)).astype(np.int32),
            "cipid": r.integers(0, 42, n_p).astype(np.int32),
            "parent": np.sort(r.integers(0, n_w, n_p)).astype(np.int32),
            "phone_al": r.integers(-500, 500, (n_p, 3)).astype(np.int32),
            "state_al": r.integers(-500, 500, (3 * n_p, 3)).astype(np.int32)}


@pytest.mark.timeout(120)
def test_gather_text_alignments_world2():
    """alignments from text have data-dependent sizes (fillers, alternates) and may be missing:
    lengths first, then one padded all_gather; every rank ends with all of them in global order"""
    n_utts = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_text_worker, args=(r, 2, port, n_utts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == n_utts
    for u in range(n_utts):
        want = _fake_text_alignment(u)
        if want is None:
            assert got[u] is None
        else:
            assert {k: v.tolist() for k, v in want.items()} == got[u]


# ---- a scaled-down config-5 job with real per-rank compute (the CPU oracle stands in for the
# GPU kernels; sharding plan, gather and CRC are the product code of soundswallower_amd.jobs) ----
_JOB = dict(n_utts=7, n_frames=48, n_phones=6)


def _oracle_job_states(utts):
    from oracle import oracle as O
    from soundswallower_amd.synth import read_raw_means, synth_alignment_task, synth_features
    mdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "soundswallower_amd", "model", "en-us")
    om = O.Model(mdir)
    means = read_raw_means(mdir)
    out = []
    for u in utts:
        scr = om.ptm_score_utt(synth_features(means, _JOB["n_frames"], 12345 + u))
        senid, tmat, _ = synth_alignment_task(om.sseq, om.phone_ssid, om.phone_tmat, om.n_ciphone,
                                              _JOB["n_phones"], 777 + u)
        rv, st, _ = om.state_align(scr, senid, tmat)
        assert rv == 0
        out.append(np.asarray(st, np.int32))
    return out


def _job_worker(rank, world, port, q):
    import torch.distributed as dist
    from soundswallower_amd.jobs import alignment_crc
    from soundswallower_amd.parallel import gather_alignments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [_JOB["n_frames"]] * _JOB["n_utts"]
    mine = shard_utterances(lens, world)[rank]
    local = _oracle_job_states(mine)
    full = gather_alignments(local, [3 * _JOB["n_phones"]] * _JOB["n_utts"], world, rank,
                             n_frames_per_utt=lens)
    if rank == 0:
        q.put((alignment_crc(full), [int(a[:, 1].sum()) for a in full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_alignment_job_crc_does_not_depend_on_rank_count():
    from soundswallower_amd.jobs import alignment_crc
    want = alignment_crc(_oracle_job_states(range(_JOB["n_utts"])))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    crc, durs = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert crc == want
    assert durs == [_JOB["n_frames"]] * _JOB["n_utts"]   # every alignment tiles its utterance
