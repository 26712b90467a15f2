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
