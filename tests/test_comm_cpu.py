"""The in-library exchange (csrc/scan_comm.cpp) without a GPU, without torch, without gloo: the ranks of a
communicator are threads of this process, the transport is a host callback over a shared buffer, and everything
between the rank's own lists and the merged answer -- record packing, all-gather, consider()'s replay over the
union -- runs through the C ABI (szg_comm_create_host, szg_comm_merge_topk, szg_comm_merge_radius).  The per-rank
scan is played by the oracle."""
import threading

import numpy as np
import pytest

import oracle as orc
from syzgydb_amd.sharded import Comm, shard_range
from syzgydb_amd import _lib


class ThreadFabric:
    """all-gather between `world` threads: everybody writes its slice, a barrier, everybody reads everything."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.buf = None
        self.lock = threading.Lock()
        self.calls = 0

    def allgather_for(self, rank):
        def allgather(send, recv):
            n = len(send)
            with self.lock:
                if self.buf is None or len(self.buf) != n * self.world:
                    self.buf = bytearray(n * self.world)
            self.barrier.wait()
            self.buf[rank * n:(rank + 1) * n] = send
            self.barrier.wait()
            recv[:] = self.buf
            self.barrier.wait()
            if rank == 0:
                self.buf = None
                self.calls += 1
            self.barrier.wait()
        return allgather


def run_ranks(world, body):
    fabric = ThreadFabric(world)
    errs, outs = [], [None] * world

    def work(rank):
        try:
            comm = Comm.host(fabric.allgather_for(rank), rank, world)
            outs[rank] = body(rank, comm)
            comm.close()
        except BaseException as e:  # noqa: B902 -- a failing rank must not leave the others in the barrier
            errs.append(e)
            fabric.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs
    return outs, fabric


@pytest.mark.parametrize("world", [1, 2, 5])
@pytest.mark.parametrize("bits,metric", [(8, 1), (32, 0)])
def test_topk_records_through_the_library(world, bits, metric):
    dim, n, k = 24, 3000, 10
    rows = orc.synth_rows(21, 0, n, dim, bits)
    Q = orc.synth_vectors(22, 0, 9, dim)

    def body(rank, comm):
        lo, hi = shard_range(n, rank, world)
        kk = k + 1
        R = np.full((len(Q), kk), np.iinfo(np.uint64).max, np.uint64)
        D = np.zeros((len(Q), kk))
        C = np.zeros(len(Q), np.int32)
        for i in range(len(Q)):
            r, d, _ = orc.search_exact(rows[lo:hi], dim, bits, metric, Q[i], k=kk)
            R[i, :len(r)] = r + np.uint64(lo)
            D[i, :len(r)] = d
            C[i] = len(r)
        comm.reserve(len(Q), k)
        out = comm.merge_topk(k, R, D, C)
        st = comm.stats()
        assert st["exchanges"] == 1 and st["rccl_ranks"] == 0
        return out
    outs, fabric = run_ranks(world, body)
    assert fabric.calls == 1                      # ONE all-gather for the batch
    for rank in range(world):
        r, d, c, hist = outs[rank]                # every rank holds the single-collection answer
        for i in range(len(Q)):
            er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=k)
            assert not hist[i]
            assert [int(x) for x in r[i, :c[i]]] == [int(x) for x in er], (rank, i)
            assert (d[i, :c[i]] == ed).all()


@pytest.mark.parametrize("world", [2, 3])
def test_radius_records_and_ties_through_the_library(world):
    # 4-bit dim-2 corpus: many equal distances; the merged order must be the reference's (heap history over the
    # union in visit order), and ranks with no hits at all take part in the padded gather
    dim, bits, metric, n = 2, 4, 0, 700
    rows = orc.synth_rows(23, 0, n, dim, bits)
    Q = orc.synth_vectors(24, 0, 5, dim)
    radii = [0.8, 0.05, 1e-9, 0.3, 2.9]

    def body(rank, comm):
        lo, hi = shard_range(n, rank, world)
        hits = []
        for i in range(len(Q)):
            r, d, _ = orc.search_exact(rows[lo:hi], dim, bits, metric, Q[i], radius=radii[i])
            hits.append((r + np.uint64(lo), d))
        return comm.merge_radius(hits)
    outs, _ = run_ranks(world, body)
    for rank in range(world):
        for i in range(len(Q)):
            er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], radius=radii[i])
            r, d = outs[rank][i]
            assert [int(x) for x in r] == [int(x) for x in er], (rank, i, len(r), len(er))
            assert (d == ed).all()
    assert any(len(outs[0][i][0]) > 50 for i in range(len(Q)))


def test_history_dependent_flag_and_short_lists():
    # equal distances at the k boundary across shards: flagged; a shard with fewer than k+1 rows: counts honoured
    world, k = 2, 3

    def body(rank, comm):
        kk = k + 1
        R = np.full((2, kk), np.iinfo(np.uint64).max, np.uint64)
        D = np.zeros((2, kk))
        C = np.zeros(2, np.int32)
        if rank == 0:
            R[0, :4], D[0, :4], C[0] = [0, 1, 2, 3], [0.1, 0.2, 0.3, 0.9], 4
            R[1, :1], D[1, :1], C[1] = [5], [0.5], 1
        else:
            R[0, :4], D[0, :4], C[0] = [64, 65, 66, 67], [0.15, 0.2, 0.8, 0.95], 4   # 0.2 twice among the best k+1 = 4
            R[1, :2], D[1, :2], C[1] = [70, 71], [0.4, 0.6], 2
        return comm.merge_topk(k, R, D, C)
    outs, _ = run_ranks(world, body)
    r, d, c, hist = outs[1]
    assert hist[0] and not hist[1]
    assert list(d[0]) == [0.1, 0.15, 0.2]
    assert c[1] == 3 and [int(x) for x in r[1]] == [70, 5, 71] and list(d[1]) == [0.4, 0.5, 0.6]


def test_callback_failure_is_an_error_not_a_crash():
    def bad(send, recv):
        raise RuntimeError("fabric down")
    comm = Comm.host(bad, 0, 1)
    R = np.zeros((1, 2), np.uint64)
    with pytest.raises(_lib.SzgError):
        comm.merge_topk(1, R, np.zeros((1, 2)), np.ones(1, np.int32))
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_that_cannot_allocate_its_staging_fails_every_rank_and_hangs_none(world):
    """ONE rank's staging growth fails (injected, as an allocation failure on one GPU of a job would): the ranks
    agree on their staging before the data exchange, so EVERY rank returns an error for that call and nobody is left
    waiting in an all-gather; the communicator stays usable and the next call -- allocation working again --
    answers correctly."""
    k, nq = 10, 400          # 400 records of 23 words: beyond the 4096 words every communicator starts with
    rng = np.random.default_rng(5)
    D = np.sort(rng.random((world, nq, k + 1)), axis=2)

    def body(rank, comm):
        R = (np.arange(nq * (k + 1), dtype=np.uint64).reshape(nq, k + 1) + np.uint64(rank * 10 ** 6))
        C = np.full(nq, k + 1, np.int32)
        if rank == world - 1:
            comm.debug_inject(1, 1)
        with pytest.raises(_lib.SzgError) as e:
            comm.merge_topk(k, R, D[rank], C)
        first = (e.value.code, str(e.value))
        out = comm.merge_topk(k, R, D[rank], C)     # the injection is spent: growth succeeds, the ranks agree
        return first, out
    outs, fabric = run_ranks(world, body)
    for rank in range(world):
        (code, text), (r, d, c, hist) = outs[rank]
        assert code == _lib.SZG_E_NOMEM, (rank, code, text)
        assert ("injected" in text) == (rank == world - 1), (rank, text)
        want = np.sort(D.transpose(1, 0, 2).reshape(nq, -1), axis=1)[:, :k]
        assert (d == want).all()
    # status round of the failed call, status round + data exchange of the good one: nothing else
    assert fabric.calls == 3


def test_uneven_radius_hits_beyond_a_small_callers_buffer():
    """rank 0 has no hits (a 4 096-entry buffer), rank 1 has 5 000 -- the merged total does not fit rank
    0's buffer.  That is rank 0's own affair: the answer is fetched again locally (szg_comm_last_radius); the ranks'
    all-gathers stay paired (exactly two for the call) and both return the same 5 000 hits."""
    world, n_hits = 2, 5000

    def body(rank, comm):
        if rank == 0:
            hits = [(np.zeros(0, np.uint64), np.zeros(0)), (np.zeros(0, np.uint64), np.zeros(0))]
        else:
            rows = np.arange(n_hits, dtype=np.uint64) + np.uint64(64)
            hits = [(rows, np.linspace(0.1, 0.9, n_hits)), (rows[:3], np.array([0.3, 0.2, 0.1]))]
        return comm.merge_radius(hits)
    outs, fabric = run_ranks(world, body)
    assert fabric.calls == 2 + 1      # counts + records, + the status round of the staging growth (10 000 words)
    for rank in range(world):
        (r0, d0), (r1, d1) = outs[rank]
        assert len(r0) == n_hits and (np.diff(d0) >= 0).all()
        assert [int(x) for x in r1] == [66, 65, 64] and list(d1) == [0.1, 0.2, 0.3]


@pytest.mark.parametrize("world", [2, 4])
def test_equal_distances_across_shards_take_the_references_order(world):
    """2-dimensional 4-bit rows: a handful of distinct distances, so the best k+1 of every query hold ties that span
    the shards.  The merge flags them, and the rank-to-rank heap chain (each rank continuing container/heap's array
    over its own rows in visit order) returns exactly what the reference's single loop returns -- ids in ITS order."""
    dim, bits, metric, n, k = 2, 4, 0, 900, 7
    rows = orc.synth_rows(31, 0, n, dim, bits)
    Q = orc.synth_vectors(32, 0, 6, dim)

    def body(rank, comm):
        lo, hi = shard_range(n, rank, world)
        kk = k + 1
        R = np.full((len(Q), kk), np.iinfo(np.uint64).max, np.uint64)
        D = np.zeros((len(Q), kk))
        C = np.zeros(len(Q), np.int32)
        for i in range(len(Q)):
            r, d, _ = orc.search_exact(rows[lo:hi], dim, bits, metric, Q[i], k=kk)
            R[i, :len(r)], D[i, :len(r)], C[i] = r + np.uint64(lo), d, len(r)
        r, d, c, hist = comm.merge_topk(k, R, D, C)
        flagged = [i for i in range(len(Q)) if hist[i]]
        dist = {i: orc.all_distances(rows[lo:hi], dim, bits, metric, Q[i]) for i in flagged}

        def replay(j, heap):    # consider()'s top-k branch (collection.go:606-619) over this rank's rows, in order
            h = orc.GoHeap(heap)
            for x, dd in enumerate(dist[flagged[j]]):
                h.consider_topk(lo + x, float(dd), k)
            return h.items()
        cr, cd, cc = comm.chain_topk(k, len(flagged), replay)
        for j, i in enumerate(flagged):
            r[i], d[i], c[i] = cr[j], cd[j], cc[j]
        return r, d, c, flagged
    outs, _ = run_ranks(world, body)
    assert len(outs[0][3]) >= 3                     # ties are the rule on this corpus
    for rank in range(world):
        r, d, c, flagged = outs[rank]
        for i in range(len(Q)):
            er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=k)
            assert [int(x) for x in r[i, :c[i]]] == [int(x) for x in er], (rank, i, i in flagged)
            assert (d[i, :c[i]] == ed).all()
