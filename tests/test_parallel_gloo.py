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
    full = gather_alignments(local, mine, n_states, world, rank)
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
