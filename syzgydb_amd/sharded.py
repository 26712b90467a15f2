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


def merge_topk_records(k, records, kk):
    """Merge straight from the all-gathered buffer: records int64 [G, nq, 2*kk+1]
    (kk rows | kk float64 bit patterns | count per query), one C call, no repacking."""
    L = _lib.load()
    G, nq, w = records.shape
    assert w == 2 * kk + 1 and records.dtype == np.int64 and records.flags.c_contiguous
    out_rows = np.empty((nq, k), dtype=np.uint64)
    out_dist = np.empty((nq, k), dtype=np.float64)
    out_count = np.empty(nq, dtype=np.int32)
    hist = np.empty(nq, dtype=np.uint8)
    _lib.check(L.szg_merge_topk_records(
        int(k), int(G), int(kk), int(nq),
        records.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
        out_rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
        out_dist.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        out_count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        hist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "szg_merge_topk_records")
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
        self._bufs = {}       # (nq, kk) -> staging of one exchange, reused call after call
        self.exchange_s = 0.0     # wall time spent in exchange() (collective + merge)
        self.exchange_host_s = 0.0  # of which: packing and merging on the host
        self.exchanges = 0

    def reset_timers(self):
        self.exchange_s = self.exchange_host_s = 0.0
        self.exchanges = 0

    def search(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        # one extra per shard so equal distances at the k boundary are visible
        return self.exchange(self.local_search(q, k + 1), k)

    def _staging(self, nq, kk):
        """Pinned host record buffers (+ device twins for RCCL) for one (nq, kk) shape."""
        import torch
        b = self._bufs.get((nq, kk))
        if b is None:
            w = 2 * kk + 1
            pin = self.device is not None
            mine_h = torch.empty((nq, w), dtype=torch.int64, pin_memory=pin)
            all_h = torch.empty((self.world * nq, w), dtype=torch.int64, pin_memory=pin)
            b = {"mine_h": mine_h, "mine_np": mine_h.numpy(), "all_h": all_h,
                 "all_np": all_h.numpy().reshape(self.world, nq, w)}
            if self.device is not None:
                b["mine_d"] = torch.empty((nq, w), dtype=torch.int64, device=self.device)
                b["all_d"] = torch.empty((self.world * nq, w), dtype=torch.int64, device=self.device)
            if len(self._bufs) > 8:
                self._bufs.clear()
            self._bufs[(nq, kk)] = b
        return b

    def exchange(self, local, k):
        """All-gather the ranks' local (rows, dist, count) top-(k+1) lists and merge."""
        import time
        import torch
        t0 = time.perf_counter()
        rows, dist, count = local
        nq, kk = rows.shape
        b = self._staging(nq, kk)
        # one int64 record per query: kk rows | kk distance bit patterns | count
        rec = b["mine_np"]
        rec[:, :kk] = rows.view(np.int64)
        rec[:, kk:2 * kk] = dist.view(np.int64)
        rec[:, 2 * kk] = count
        t1 = time.perf_counter()
        # concatenated along dim 0 (the form both RCCL and gloo accept), viewed [world, nq, .]
        if self.device is not None:
            b["mine_d"].copy_(b["mine_h"], non_blocking=True)
            self._dist.all_gather_into_tensor(b["all_d"], b["mine_d"], group=self.group)
            b["all_h"].copy_(b["all_d"], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        else:
            self._dist.all_gather_into_tensor(b["all_h"], b["mine_h"], group=self.group)
        t2 = time.perf_counter()
        out = merge_topk_records(k, b["all_np"], kk)
        t3 = time.perf_counter()
        self.exchange_s += t3 - t0
        self.exchange_host_s += (t1 - t0) + (t3 - t2)
        self.exchanges += 1
        return out

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

    def search_radius_stream(self, local_radius, queries, radius):
        """Radius searches back to back: the rank's sweep for query i+1 runs in a worker thread
        while this thread exchanges and merges query i (every rank calls with the same queries)."""
        from concurrent.futures import ThreadPoolExecutor
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, queries.shape[-1])
        outs = []
        with ThreadPoolExecutor(max_workers=1) as ex:
            futs = [ex.submit(local_radius, q[i], float(radius)) for i in range(q.shape[0])]
            for i, f in enumerate(futs):
                res = f.result()
                outs.append(self.search_radius(lambda _q, _r, res=res: res, q[i], radius))
        return outs

    def search_stream(self, queries, k, chunk):
        """Pipelined form for throughput: the local sweeps of chunk i+1 run in a worker
        thread (the C call releases the GIL) while this thread exchanges and merges
        chunk i.  Every rank must call it with the same chunking."""
        q = np.ascontiguousarray(queries, dtype=np.float64)
        chunks = [q[i:i + chunk] for i in range(0, q.shape[0], chunk)]
        if len(chunks) == 1:   # nothing to overlap: no worker thread
            return self.exchange(self.local_search(chunks[0], k + 1), k)
        from concurrent.futures import ThreadPoolExecutor
        outs = []
        with ThreadPoolExecutor(max_workers=1) as ex:
            futs = [ex.submit(self.local_search, c, k + 1) for c in chunks]
            for f in futs:
                outs.append(self.exchange(f.result(), k))
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(4))
