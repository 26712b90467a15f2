"""One-process-per-GPU sharding of the exact scan: thin callers of the library.

Rows are split into contiguous ranges, one per rank; every rank answers each query on its own
range through the C ABI (exact local top-(k+1), rows made global with szg_index_set_row_base).
The exchange lives in libsyzgy_scan.so (csrc/scan_comm.cpp): ONE all-gather per micro-batch of
int64 records -- ncclAllGather (RCCL over xGMI) on a communicator the library creates itself, or
a host callback (the tests: gloo on CPU, several ranks on one card) -- followed by the
reference's selection (collection.go:606-619) replayed over the union on every rank.

torch.distributed is plumbing here: rendezvous, handing the 128-byte communicator id round, and
the CPU transport of the tests.  A Go or C++ host does the same with its own means
(include/syzgy_scan.h, "one process per GPU").
"""
import ctypes

import numpy as np

from . import _lib


def shard_range(n_rows, rank, world):
    """Contiguous row range of `rank`; boundaries are multiples of 64 (mask words)."""
    per = (n_rows + world - 1) // world
    per = (per + 63) // 64 * 64
    lo = min(rank * per, n_rows)
    hi = min(lo + per, n_rows)
    return lo, hi


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def merge_topk(k, rows, dist, counts):
    """Merge per-shard results held by ONE process.  rows/dist: [G, nq, L], counts: [G, nq].

    Returns (rows[nq,k] uint64, dist[nq,k] float64, count[nq] int32, history_dependent[nq] bool).
    """
    L = _lib.load()
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    G, nq, ll = rows.shape
    out_rows = np.zeros((nq, k), dtype=np.uint64)
    out_dist = np.zeros((nq, k), dtype=np.float64)
    out_count = np.zeros(nq, dtype=np.int32)
    hist = np.zeros(nq, dtype=np.uint8)
    _lib.check(L.szg_merge_topk(
        int(k), int(G), int(ll), int(nq), _p(rows, ctypes.c_uint64), _p(dist, ctypes.c_double),
        _p(counts, ctypes.c_int32), _p(out_rows, ctypes.c_uint64), _p(out_dist, ctypes.c_double),
        _p(out_count, ctypes.c_int32), _p(hist, ctypes.c_uint8)), "szg_merge_topk")
    return out_rows, out_dist, out_count, hist.astype(bool)


def merge_topk_records(k, records, kk):
    """Merge straight from an all-gathered buffer: records int64 [G, nq, 2*kk+1]
    (kk rows | kk float64 bit patterns | count per query), one C call, no repacking."""
    L = _lib.load()
    G, nq, w = records.shape
    assert w == 2 * kk + 1 and records.dtype == np.int64 and records.flags.c_contiguous
    out_rows = np.empty((nq, k), dtype=np.uint64)
    out_dist = np.empty((nq, k), dtype=np.float64)
    out_count = np.empty(nq, dtype=np.int32)
    hist = np.empty(nq, dtype=np.uint8)
    _lib.check(L.szg_merge_topk_records(
        int(k), int(G), int(kk), int(nq), _p(records, ctypes.c_int64), _p(out_rows, ctypes.c_uint64),
        _p(out_dist, ctypes.c_double), _p(out_count, ctypes.c_int32), _p(hist, ctypes.c_uint8)),
        "szg_merge_topk_records")
    return out_rows, out_dist, out_count, hist.astype(bool)


class Comm:
    """szg_comm: the communicator of the sharded searches (include/syzgy_scan.h)."""

    def __init__(self, handle, rank, world, keep=None):
        self._L = _lib.load()
        self._h = handle
        self.rank, self.world = int(rank), int(world)
        self._keep = keep  # the callback trampoline of a host transport

    @staticmethod
    def unique_id():
        """128 bytes rank 0 creates and hands to the other ranks (any host-side means)."""
        L = _lib.load()
        buf = (ctypes.c_uint8 * _lib.SZG_COMM_ID_BYTES)()
        _lib.check(L.szg_comm_unique_id(buf), "szg_comm_unique_id")
        return bytes(buf)

    @classmethod
    def rccl(cls, comm_id, rank, world, device):
        """Collective: ncclCommInitRank on `device` inside the library (+ one untimed all-gather)."""
        L = _lib.load()
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * _lib.SZG_COMM_ID_BYTES).from_buffer_copy(comm_id)
        _lib.check(L.szg_comm_create(ctypes.byref(h), buf, int(rank), int(world), int(device)), "szg_comm_create")
        return cls(h, rank, world)

    @classmethod
    def host(cls, allgather, rank, world):
        """The host's own transport: allgather(send: memoryview, recv: memoryview) fills recv
        ([world][len(send)] bytes, rank order) from every rank's send.  No device involved."""
        L = _lib.load()
        world = int(world)

        def thunk(_user, send, recv, nbytes):
            try:
                n = int(nbytes)
                s = (ctypes.c_uint8 * n).from_address(send)
                r = (ctypes.c_uint8 * (n * world)).from_address(recv)
                allgather(memoryview(s).cast("B"), memoryview(r).cast("B"))
                return 0
            except Exception:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        cb = _lib.ALLGATHER_FN(thunk)
        h = ctypes.c_void_p()
        _lib.check(L.szg_comm_create_host(ctypes.byref(h), cb, None, int(rank), world), "szg_comm_create_host")
        return cls(h, rank, world, keep=cb)

    @classmethod
    def from_process_group(cls, group=None, device=None, via_torch_device=None):
        """A communicator for the ranks of a torch.distributed group.  device = GPU ordinal: the library's own
        RCCL communicator (the id travels through the group once).  Otherwise the group's own all-gather as host
        transport: on the CPU (gloo: tests and one-card rehearsals), or through `via_torch_device` for an "nccl"
        group (fallback when the library cannot create its communicator)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if device is not None:
            box = [cls.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            return cls.rccl(box[0], rank, world, device)

        def allgather(send, recv):
            s = torch.frombuffer(send, dtype=torch.uint8)
            r = torch.frombuffer(recv, dtype=torch.uint8)
            if via_torch_device is None:
                dist.all_gather_into_tensor(r, s, group=group)
            else:
                rd = torch.empty(r.numel(), dtype=torch.uint8, device=via_torch_device)
                dist.all_gather_into_tensor(rd, s.to(via_torch_device), group=group)
                r.copy_(rd)
        return cls.host(allgather, rank, world)

    def close(self):
        if self._h:
            self._L.szg_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, n_queries, k):
        _lib.check(self._L.szg_comm_reserve(self._h, int(n_queries), int(k)), "szg_comm_reserve")

    def stats(self):
        s = _lib.SzgCommStats()
        _lib.check(self._L.szg_comm_get_stats(self._h, ctypes.byref(s)), "szg_comm_get_stats")
        return {name: getattr(s, name) for name, _ in _lib.SzgCommStats._fields_}

    def reset_stats(self):
        _lib.check(self._L.szg_comm_reset_stats(self._h), "szg_comm_reset_stats")

    def merge_topk(self, k, rows, dist, counts):
        """Collective: this rank's exact top-(k+1) lists (rows GLOBAL [nq,k+1], dist, counts [nq]) -> the
        single-collection top-k on every rank: (rows [nq,k], dist, count, history_dependent)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        nq, kk = rows.shape
        if kk != k + 1 or dist.shape != rows.shape or counts.shape != (nq,):
            raise ValueError("local lists must be [n_queries, k+1]")
        out_rows = np.empty((nq, k), dtype=np.uint64)
        out_dist = np.empty((nq, k), dtype=np.float64)
        out_count = np.empty(nq, dtype=np.int32)
        hist = np.empty(nq, dtype=np.uint8)
        _lib.check(self._L.szg_comm_merge_topk(
            self._h, int(k), int(nq), _p(rows, ctypes.c_uint64), _p(dist, ctypes.c_double), _p(counts, ctypes.c_int32),
            _p(out_rows, ctypes.c_uint64), _p(out_dist, ctypes.c_double), _p(out_count, ctypes.c_int32),
            _p(hist, ctypes.c_uint8)), "szg_comm_merge_topk")
        return out_rows, out_dist, out_count, hist.astype(bool)

    def merge_radius(self, hits):
        """Collective: this rank's radius hits, a list of (rows GLOBAL, dist) per query -> the merged list."""
        nq = len(hits)
        off = np.zeros(nq + 1, dtype=np.uint64)
        for i, (r, _) in enumerate(hits):
            off[i + 1] = off[i] + len(r)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint64) for r, _ in hits] or
                                                   [np.zeros(0, np.uint64)]))
        dist = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.float64) for _, d in hits] or
                                                   [np.zeros(0, np.float64)]))
        # The buffer's size is this rank's own business: a too-small one makes the call return SZG_E_TRUNCATED with
        # the offsets complete, and the answer -- kept by the communicator -- is fetched again LOCALLY.  (Repeating
        # the collective on the truncated ranks only would pair their all-gathers with the other ranks' next ones.)
        cap = max(1 << 12, 4 * int(off[nq]))
        out_rows = np.zeros(cap, dtype=np.uint64)
        out_dist = np.zeros(cap, dtype=np.float64)
        out_off = np.zeros(nq + 1, dtype=np.uint64)
        rc = self._L.szg_comm_merge_radius(
            self._h, nq, _p(off, ctypes.c_uint64), _p(rows, ctypes.c_uint64) if rows.size else None,
            _p(dist, ctypes.c_double) if dist.size else None, _p(out_rows, ctypes.c_uint64),
            _p(out_dist, ctypes.c_double), cap, _p(out_off, ctypes.c_uint64))
        if rc == _lib.SZG_E_TRUNCATED:
            cap = int(out_off[nq])
            out_rows = np.zeros(cap, dtype=np.uint64)
            out_dist = np.zeros(cap, dtype=np.float64)
            rc = self._L.szg_comm_last_radius(self._h, nq, _p(out_rows, ctypes.c_uint64), _p(out_dist, ctypes.c_double),
                                              cap, _p(out_off, ctypes.c_uint64))
        _lib.check(rc, "szg_comm_merge_radius")
        return [(out_rows[int(out_off[i]):int(out_off[i + 1])], out_dist[int(out_off[i]):int(out_off[i + 1])])
                for i in range(nq)]

    def chain_topk(self, k, n_flagged, replay):
        """Collective: the rank-to-rank heap chain for n_flagged queries whose merged answer held equal distances.
        replay(j, heap) -> heap continues the reference's heap -- a list of (row, distance) in container/heap's array
        order -- over THIS rank's rows of flagged query j in visit order.  Returns (rows [n,k], dist, count)."""
        def thunk(_user, j, kk, hrows, hdist, hn):
            try:
                heap = [(int(hrows[i]), float(hdist[i])) for i in range(hn[0])]
                heap = replay(int(j), heap)
                if len(heap) > kk:
                    return 1
                for i, (r, d) in enumerate(heap):
                    hrows[i], hdist[i] = r, d
                hn[0] = len(heap)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = _lib.REPLAY_FN(thunk)
        out_rows = np.empty((n_flagged, k), dtype=np.uint64)
        out_dist = np.empty((n_flagged, k), dtype=np.float64)
        out_count = np.empty(n_flagged, dtype=np.int32)
        _lib.check(self._L.szg_comm_chain_topk(self._h, int(k), int(n_flagged), cb, None, _p(out_rows, ctypes.c_uint64),
                                               _p(out_dist, ctypes.c_double), _p(out_count, ctypes.c_int32)),
                   "szg_comm_chain_topk")
        return out_rows, out_dist, out_count

    def debug_inject(self, what, value):
        _lib.check(self._L.szg_comm_debug_inject(self._h, int(what), int(value)), "szg_comm_debug_inject")


class ShardedSearcher:
    """Exact top-k / radius search over a corpus sharded across the ranks of a communicator.

    With a ScanIndex (`index=`) everything happens inside the library: szg_search_topk_sharded /
    szg_search_radius_sharded (local sweeps, exchange, merge; long query lists pipelined by a worker thread).
    With a callable local_search(queries[nq,dim], kk) -> (rows[nq,kk] uint64 GLOBAL, dist, count) -- the CPU tests,
    where the oracle plays the per-rank scan -- the library does the exchange and the merge (szg_comm_merge_*).
    """

    def __init__(self, local_search=None, group=None, device=None, index=None, comm=None):
        self.local_search = local_search
        self.index = index
        self.comm = comm if comm is not None else Comm.from_process_group(group, device)
        self.rank, self.world = self.comm.rank, self.comm.world
        if index is not None:
            index.attach_comm(self.comm)

    def close(self):
        if self.index is not None:
            self.index.attach_comm(None)
        self.comm.close()

    # -- top-k ---------------------------------------------------------------
    def search(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if self.index is not None:
            return self.index.search_topk_sharded(q, k)
        # one extra per shard so equal distances at the k boundary are visible
        return self.comm.merge_topk(k, *self.local_search(q, k + 1))

    def search_stream(self, queries, k, chunk=None):
        """Throughput form.  With an index the library pipelines by itself (micro-batches of 256 behind a worker
        thread); with a callable the chunks go one after the other.  Every rank must use the same chunking."""
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if self.index is not None or not chunk or chunk >= q.shape[0]:
            return self.search(q, k)
        outs = [self.search(q[i:i + chunk], k) for i in range(0, q.shape[0], chunk)]
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(4))

    # -- radius (config #5's mode) -------------------------------------------------
    def search_radius(self, local_radius, query, radius):
        """One radius search; local_radius(query, radius) -> (rows uint64 GLOBAL, dist) is the rank's own
        (ignored with an index).  Returns the single-collection answer, ties included."""
        q = np.ascontiguousarray(query, dtype=np.float64).reshape(1, -1)
        if self.index is not None:
            return self.index.search_radius_sharded(q, float(radius))[0]
        return self.comm.merge_radius([local_radius(q[0], float(radius))])[0]

    def search_radius_batch(self, queries, radii, local_radius=None):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        q = q.reshape(-1, q.shape[-1])
        rad = np.broadcast_to(np.asarray(radii, dtype=np.float64), (q.shape[0],))
        if self.index is not None:
            return self.index.search_radius_sharded(q, rad)
        return self.comm.merge_radius([local_radius(q[i], float(rad[i])) for i in range(q.shape[0])])
