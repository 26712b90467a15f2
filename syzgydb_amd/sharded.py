"""One-process-per-GPU sharding of the exact scan.

Rows are split into contiguous ranges, one per rank; every rank answers each
query on its own range through the C ABI (exact local top-(k+1), rows made
global with szg_index_set_row_base), the per-rank lists are exchanged with ONE
all-gather per query batch (RCCL over xGMI when the process group is "nccl";
gloo on CPU in the tests) and merged on every rank by szg_merge_topk, which
replays the reference's selection (collection.go:606-619) over the union.

torch.distributed is plumbing here: rendezvous, the collective, barriers.
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


def merge_topk(k, rows, dist, counts):
    """Merge per-shard results.  rows/dist: [G, nq, L], counts: [G, nq].

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
        int(k), int(G), int(ll), int(nq),
        rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
        dist.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        out_rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
        out_dist.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        out_count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        hist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "szg_merge_topk")
    return out_rows, out_dist, out_count, hist.astype(bool)


class ShardedSearcher:
    """Exact top-k over a corpus sharded across the ranks of a process group.

    local_search(queries[nq,dim], kk) -> (rows[nq,kk] uint64 GLOBAL, dist[nq,kk], count[nq])
    is the rank's own scan (ScanIndex.search_topk on a handle with row_base set).
    """

    def __init__(self, local_search, group=None, device=None):
        import torch.distributed as dist
        self._dist = dist
        self.local_search = local_search
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device  # torch device of the exchange buffers ("cuda:N" for nccl)

    def search(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        # one extra per shard so equal distances at the k boundary are visible
        return self.exchange(self.local_search(q, k + 1), k)

    def exchange(self, local, k):
        """All-gather the ranks' local (rows, dist, count) top-(k+1) lists and merge."""
        import torch
        rows, dist, count = local
        nq, kk = rows.shape
        # one int64 record per query: kk rows | kk distance bit patterns | count
        rec = np.zeros((nq, 2 * kk + 1), dtype=np.int64)
        rec[:, :kk] = rows.view(np.int64)
        rec[:, kk:2 * kk] = dist.view(np.int64)
        rec[:, 2 * kk] = count
        mine = torch.from_numpy(rec)
        if self.device is not None:
            mine = mine.to(self.device)
        # concatenated along dim 0 (the form both RCCL and gloo accept), viewed [world, nq, .]
        gathered = torch.empty((self.world * nq, mine.shape[1]), dtype=mine.dtype,
                               device=mine.device)
        self._dist.all_gather_into_tensor(gathered, mine, group=self.group)
        g = gathered.cpu().numpy().reshape(self.world, nq, -1)
        g_rows = np.ascontiguousarray(g[:, :, :kk]).view(np.uint64)
        g_dist = np.ascontiguousarray(g[:, :, kk:2 * kk]).view(np.float64)
        g_count = np.ascontiguousarray(g[:, :, 2 * kk]).astype(np.int32)
        return merge_topk(k, g_rows, g_dist, g_count)

    def search_radius(self, local_radius, query, radius):
        """Radius search over the sharded corpus (config #5's mode).

        local_radius(query, radius) -> (rows uint64 GLOBAL, dist float64) is the rank's own
        szg_search_radius.  The ranks all-gather their hit counts, then one padded
        all-gather of (row, distance-bits) records; every rank replays the reference's
        push-all / pop-all heap (collection.go:598-603, :694-697) over the union in row
        order, so the returned order is the single-collection one, ties included.
        """
        import torch
        q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
        rows, dist = local_radius(q, float(radius))
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        n = torch.tensor([rows.size], dtype=torch.int64)
        if self.device is not None:
            n = n.to(self.device)
        counts = torch.empty(self.world, dtype=torch.int64, device=n.device)
        self._dist.all_gather_into_tensor(counts, n, group=self.group)
        counts = counts.cpu().numpy()
        total, m = int(counts.sum()), int(counts.max())
        if total == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float64)
        rec = np.zeros((m, 2), dtype=np.int64)
        rec[:rows.size, 0] = rows.view(np.int64)
        rec[:rows.size, 1] = dist.view(np.int64)
        mine = torch.from_numpy(rec)
        if self.device is not None:
            mine = mine.to(self.device)
        gathered = torch.empty((self.world * m, 2), dtype=torch.int64, device=mine.device)
        self._dist.all_gather_into_tensor(gathered, mine, group=self.group)
        g = gathered.cpu().numpy().reshape(self.world, 1, m, 2)
        g_rows = np.ascontiguousarray(g[..., 0]).view(np.uint64)
        g_dist = np.ascontiguousarray(g[..., 1]).view(np.float64)
        r, d, c, _ = merge_topk(total, g_rows, g_dist, counts.astype(np.int32).reshape(self.world, 1))
        return r[0, :c[0]], d[0, :c[0]]

    def search_stream(self, queries, k, chunk):
        """Pipelined form for throughput: the local sweeps of chunk i+1 run in a worker
        thread (the C call releases the GIL) while this thread exchanges and merges
        chunk i.  Every rank must call it with the same chunking."""
        from concurrent.futures import ThreadPoolExecutor
        q = np.ascontiguousarray(queries, dtype=np.float64)
        chunks = [q[i:i + chunk] for i in range(0, q.shape[0], chunk)]
        outs = []
        with ThreadPoolExecutor(max_workers=1) as ex:
            futs = [ex.submit(self.local_search, c, k + 1) for c in chunks]
            for f in futs:
                outs.append(self.exchange(f.result(), k))
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(4))
