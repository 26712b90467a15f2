"""The in-library exchange on the GPU (csrc/scan_comm.cpp), through the C ABI only -- no torch in any of these
processes.

* RCCL with ONE rank on the box's card: ncclGetUniqueId / ncclCommInitRank / ncclAllGather inside the library,
  pinned + device staging, the pipelined szg_search_topk_sharded (several micro-batches) and the radius form.
* TWO processes on one card with a host transport (a multiprocessing pipe): RCCL refuses two ranks on one device, so
  this is the 2-rank rehearsal of everything but the collective itself.
* TWO processes, one card each, RCCL between them: runs where the box has >= 2 devices (the driver's 8-GPU node).
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(96, 32, 1, 20000, 10, 31), (48, 8, 0, 9000, 100, 32), (384, 4, 1, 12000, 5, 33),
         (2, 4, 0, 3000, 7, 34)]   # the last: a handful of distinct distances -- ties across the shards, the heap chain


def _device_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    n = ctypes.c_int(0)
    return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0


def _check_rank(rank, world, comm, device, n_queries=20):
    """What every rank does: shard, search through the library, compare with the oracle on the WHOLE corpus."""
    import oracle as orc
    from syzgydb_amd import ScanIndex
    from syzgydb_amd.sharded import shard_range
    for dim, bits, metric, n, k, seed in CASES:
        rows = orc.synth_rows(seed, 0, n, dim, bits)
        Q = orc.synth_vectors(seed + 100, 0, n_queries, dim)
        lo, hi = shard_range(n, rank, world)
        with ScanIndex(dim, bits, metric, devices=[device]) as ix:
            ix.synth(hi - lo, seed, first_row=lo)
            ix.set_row_base(lo)
            ix.attach_comm(comm)
            comm.reset_stats()
            r, d, c, hist = ix.search_topk_sharded(Q, k)
            expect = 1 if n_queries <= 128 else (n_queries + 255) // 256
            assert comm.stats()["exchanges"] == expect, comm.stats()   # ONE all-gather per micro-batch
            for i in range(len(Q)):
                er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=k)
                # (equal distances among the best k+1 -- hist[i] -- were settled by the rank-to-rank heap chain: the
                # order is the reference's too)
                assert [int(x) for x in r[i, :c[i]]] == [int(x) for x in er], (rank, dim, i, bool(hist[i]))
                assert (d[i, :c[i]] == ed).all()
            # radius: per-query radii from the merged top-k (the k-th distance: k hits, ties aside)
            radii = np.maximum(d[:6, k - 1], 1e-9)
            hits = ix.search_radius_sharded(Q[:6], radii)
            for i in range(6):
                er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], radius=float(radii[i]))
                assert [int(x) for x in hits[i][0]] == [int(x) for x in er], (rank, dim, i)
                assert (hits[i][1] == ed).all()
            # a filter on this rank's own rows (local mask), every third row
            mask = np.zeros((4, hi - lo), bool)
            mask[:, (np.arange(lo, hi) % 3) == 0] = True
            r, d, c, hist = ix.search_topk_sharded(Q[:4], k, allow=mask)
            allow = (np.arange(n) % 3) == 0
            for i in range(4):
                er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=k, allow=allow)
                assert [int(x) for x in r[i, :c[i]]] == [int(x) for x in er], (rank, dim, i, bool(hist[i]))
                assert (d[i, :c[i]] == ed).all()
            ix.attach_comm(None)


def test_rccl_inside_the_library_one_rank():
    sys.path.insert(0, ROOT)
    from syzgydb_amd.sharded import Comm
    comm = Comm.rccl(Comm.unique_id(), 0, 1, 0)
    st = comm.stats()
    assert st["rccl_ranks"] == 1 and st["exchanges"] == 0     # (the attach-time all-gather is not counted)
    _check_rank(0, 1, comm, 0)
    _check_rank(0, 1, comm, 0, n_queries=300)                 # pipelined: 2 micro-batches behind a worker thread
    comm.close()


def _pipe_rank(rank, conn, id_path):
    """Rank of the 2-process rehearsal: the transport is the pipe to the other rank."""
    sys.path.insert(0, ROOT)
    from syzgydb_amd.sharded import Comm

    def allgather(send, recv):
        n = len(send)
        conn.send_bytes(bytes(send))
        other = conn.recv_bytes()
        recv[rank * n:(rank + 1) * n] = send
        recv[(1 - rank) * n:(2 - rank) * n] = other
    comm = Comm.host(allgather, rank, 2)
    _check_rank(rank, 2, comm, 0)
    comm.close()


def _rccl_rank(rank, conn, id_path):
    """Rank of the real thing: one card per process, the id travels through a file."""
    sys.path.insert(0, ROOT)
    from syzgydb_amd.sharded import Comm
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(Comm.unique_id())
        os.rename(id_path + ".tmp", id_path)
    t0 = time.time()
    while not os.path.exists(id_path):
        assert time.time() - t0 < 60, "rank 0 never published the communicator id"
        time.sleep(0.05)
    comm = Comm.rccl(open(id_path, "rb").read(), rank, 2, rank)
    assert comm.stats()["rccl_ranks"] == 2
    _check_rank(rank, 2, comm, rank)
    _check_rank(rank, 2, comm, rank, n_queries=300)
    comm.close()


def _two_processes(target, tmp_path):
    ctx = mp.get_context("spawn")   # children start clean: no GPU state inherited from the test process
    a, b = ctx.Pipe()
    id_path = str(tmp_path / "comm.id")
    ps = [ctx.Process(target=target, args=(0, a, id_path)), ctx.Process(target=target, args=(1, b, id_path))]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
    for p in ps:
        if p.is_alive():
            p.terminate()
    assert [p.exitcode for p in ps] == [0, 0]


def test_two_ranks_on_one_card_host_transport_c_abi_only(tmp_path):
    _two_processes(_pipe_rank, tmp_path)


@pytest.mark.skipif(_device_count() < 2, reason="RCCL wants one device per rank: needs >= 2 GPUs")
def test_two_ranks_two_cards_rccl_c_abi_only(tmp_path):
    _two_processes(_rccl_rank, tmp_path)
