"""ssw_gather_alignments with more than one rank, on the CPU.

The C entry point pads every rank's entries to the largest count, exchanges them with ONE
all-gather and packs rank r's share back to back into the output.  Until round 3 that code had
only ever run on a one-rank communicator.  ssw_comm_from_transport lets the host supply the
all-gather (what an MPI host would do with MPI_Allgather); here it is a test double that joins
the ranks -- threads of this process -- at a barrier, so the padding, the per-rank offsets and
the unpacking run for 2, 3 and 5 ranks with ragged counts, an empty rank included, and the
error paths (a count that contradicts what the rank passes, a failing transport) are checked."""
import threading

import numpy as np
import pytest

from soundswallower_amd import _lib
from soundswallower_amd.parallel import TransportComm, gather_alignments, shard_utterances


class FakeWorld:
    """An all-gather among `n` threads of one process."""

    def __init__(self, n, fail_on=None):
        self.n, self.fail_on = n, fail_on
        self.slots = [None] * n
        self.barrier = threading.Barrier(n)
        self.calls = 0

    def all_gather_for(self, rank):
        def fn(send, recv, n_words):
            assert recv.shape == (self.n, n_words)
            self.slots[rank] = send.copy()
            self.barrier.wait(timeout=30)
            for r in range(self.n):
                assert len(self.slots[r]) == n_words, "ranks disagree on the padded size"
                recv[r, :] = self.slots[r]
            self.barrier.wait(timeout=30)
            self.calls += 1
            if self.fail_on == rank:
                raise RuntimeError("transport failure")
        return fn


def _run_ranks(n, body):
    out, errs = [None] * n, [None] * n

    def work(r):
        try:
            out[r] = body(r)
        except Exception as e:      # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    return out, errs


@pytest.mark.parametrize("counts", [[5, 2], [0, 7, 3], [4, 4, 4], [1, 0, 9, 2, 6], [3]])
def test_ragged_counts_are_packed_in_rank_order(counts):
    _lib.build()
    n = len(counts)
    world = FakeWorld(n)
    rng = np.random.default_rng(7)
    local = [rng.integers(-1000, 1000, (c, 3)).astype(np.int32) for c in counts]

    def body(r):
        comm = TransportComm(world.all_gather_for(r), n, r)
        try:
            first = comm.gather(local[r], counts)
            again = comm.gather(local[r] + 1, counts)      # staging is reused
            return first, again
        finally:
            comm.close()
    out, errs = _run_ranks(n, body)
    assert errs == [None] * n, errs
    want = np.concatenate(local)
    for r in range(n):
        assert np.array_equal(out[r][0], want), r
        assert np.array_equal(out[r][1], want + 1), r


def test_gather_alignments_through_the_c_entry_point_with_three_ranks():
    """parallel.gather_alignments(comm=...) = shard plan + C gather: utterances of different
    lengths dealt over 3 ranks come back in global utterance order on every rank."""
    _lib.build()
    n_frames = [30, 10, 50, 20, 40, 60, 15]
    n_states = [9, 3, 15, 6, 12, 18, 6]
    plan = shard_utterances(n_frames, 3)
    world = FakeWorld(3)
    per_utt = [np.full((n_states[u], 3), u, np.int32) + np.arange(n_states[u])[:, None]
               for u in range(len(n_frames))]

    def body(r):
        comm = TransportComm(world.all_gather_for(r), 3, r)
        try:
            return gather_alignments([per_utt[u] for u in plan[r]], n_states, 3, r,
                                     n_frames_per_utt=n_frames, comm=comm)
        finally:
            comm.close()
    out, errs = _run_ranks(3, body)
    assert errs == [None] * 3, errs
    for r in range(3):
        assert len(out[r]) == len(n_frames)
        for u in range(len(n_frames)):
            assert np.array_equal(out[r][u], per_utt[u]), (r, u)


def test_errors_are_reported_not_swallowed():
    _lib.build()
    world = FakeWorld(1)
    comm = TransportComm(world.all_gather_for(0), 1, 0)
    with pytest.raises(RuntimeError, match="counts"):
        comm.gather(np.zeros((2, 3), np.int32), [3])           # the rank passes 2, counts say 3
    assert comm.gather(np.zeros((0, 3), np.int32), [0]).shape == (0, 3)   # nothing to exchange
    assert world.calls == 0
    comm.close()
    bad = FakeWorld(1, fail_on=0)
    comm = TransportComm(bad.all_gather_for(0), 1, 0)
    with pytest.raises(RuntimeError, match="all-gather returned"):
        comm.gather(np.ones((2, 3), np.int32), [2])
    comm.close()
    L = _lib.lib()
    assert not L.ssw_comm_from_transport(None, None, 2, 0)
    assert "bad arguments" in _lib.last_error()
