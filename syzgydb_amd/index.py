"""ScanIndex: thin Python handle over the C ABI of include/syzgy_scan.h.

One ScanIndex == one szg_index == the HBM mirror of one Collection's packed
vectors.  All compute happens in libsyzgy_scan.so (HIP, gfx950).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import SZG_COSINE, SZG_EUCLIDEAN, SzgError, SzgStats, check  # noqa: F401


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _u64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def _f64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def pack_allow_bits(mask):
    """bool[n_rows] (or [n_queries, n_rows]) -> uint64 words, bit r of word r//64."""
    m = np.atleast_2d(np.asarray(mask, dtype=bool))
    nq, n = m.shape
    words = (n + 63) // 64
    padded = np.zeros((nq, words * 64), dtype=np.uint8)
    padded[:, :n] = m
    packed = np.packbits(padded, axis=1, bitorder="little")
    return np.ascontiguousarray(packed).view(np.uint64).reshape(nq, words)


class ScanIndex:
    def __init__(self, dim, quant_bits, metric, devices=None):
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        self.dim = int(dim)
        self.quant_bits = int(quant_bits)
        self.metric = int(metric)
        dev_arr = None
        n_dev = 0
        if devices is not None:
            devices = list(devices)
            dev_arr = (ctypes.c_int * len(devices))(*devices)
            n_dev = len(devices)
        check(self._L.szg_index_create(ctypes.byref(self._h), self.dim, self.quant_bits,
                                       self.metric, dev_arr, n_dev), "szg_index_create")
        self.row_bytes = int(self._L.szg_row_bytes(self.quant_bits, self.dim))
        self.options = {}   # tunables set through this object (the host mirrors consult tie_mode)
        self._comm = None
        # SZG_OPTIONS="name=value,...": tunables applied to every new handle (test sweeps)
        for item in os.environ.get("SZG_OPTIONS", "").split(","):
            if "=" in item:
                name, value = item.split("=", 1)
                self.set_option(name.strip(), int(value))

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if self._h:
            self._L.szg_index_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- corpus ---------------------------------------------------------------
    def _rows_arg(self, rows):
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        if a.size % self.row_bytes:
            raise ValueError("rows size is not a multiple of row_bytes=%d" % self.row_bytes)
        return a, a.size // self.row_bytes

    def load(self, rows):
        a, n = self._rows_arg(rows)
        check(self._L.szg_index_load(self._h, _u8(a) if n else None, n), "szg_index_load")

    def append(self, rows):
        a, n = self._rows_arg(rows)
        check(self._L.szg_index_append(self._h, _u8(a) if n else None, n), "szg_index_append")

    def append_vectors(self, vectors):
        """Bulk AddDocument from float64 vectors: quantized and packed on the device."""
        v = np.ascontiguousarray(vectors, dtype=np.float64).reshape(-1, self.dim)
        check(self._L.szg_index_append_f64(self._h, _f64(v) if v.size else None, v.shape[0]),
              "szg_index_append_f64")

    def distances(self, query, rows):
        """The reference's float64 distance from `query` to each listed row."""
        q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
        if q.size != self.dim:
            raise ValueError("query length %d != dimension %d" % (q.size, self.dim))
        r = np.ascontiguousarray(rows, dtype=np.uint64).reshape(-1)
        out = np.zeros(r.size, dtype=np.float64)
        check(self._L.szg_distances(self._h, _f64(q), _u64(r) if r.size else None, r.size,
                                    _f64(out) if r.size else None), "szg_distances")
        return out

    def pair_distances(self, rows_a, rows_b):
        """The reference's float64 distance between stored rows a[i] and b[i]."""
        a = np.ascontiguousarray(rows_a, dtype=np.uint64).reshape(-1)
        b = np.ascontiguousarray(rows_b, dtype=np.uint64).reshape(-1)
        if a.size != b.size:
            raise ValueError("rows_a and rows_b differ in length")
        out = np.zeros(a.size, dtype=np.float64)
        if a.size:
            check(self._L.szg_pair_distances(self._h, _u64(a), _u64(b), a.size, _f64(out)),
                  "szg_pair_distances")
        return out

    def overwrite(self, row, row_bytes):
        a, n = self._rows_arg(row_bytes)
        if n != 1:
            raise ValueError("overwrite takes exactly one row")
        check(self._L.szg_index_overwrite(self._h, int(row), _u8(a)), "szg_index_overwrite")

    def overwrite_vector(self, row, vector):
        """UpdateDocument from a float64 vector: quantized and packed on the device."""
        v = np.ascontiguousarray(vector, dtype=np.float64).reshape(-1)
        if v.size != self.dim:
            raise ValueError("vector length %d != dimension %d" % (v.size, self.dim))
        check(self._L.szg_index_overwrite_f64(self._h, int(row), _f64(v)), "szg_index_overwrite_f64")

    def tombstone(self, row):
        check(self._L.szg_index_tombstone(self._h, int(row)), "szg_index_tombstone")

    def synth(self, n_rows, seed, first_row=0):
        check(self._L.szg_index_synth(self._h, int(n_rows), int(seed), int(first_row)),
              "szg_index_synth")

    def read_rows(self, first_row, n_rows):
        out = np.zeros((int(n_rows), self.row_bytes), dtype=np.uint8)
        check(self._L.szg_index_read_rows(self._h, int(first_row), int(n_rows),
                                          _u8(out) if n_rows else None), "szg_index_read_rows")
        return out

    def set_row_base(self, base):
        check(self._L.szg_index_set_row_base(self._h, int(base)), "szg_index_set_row_base")

    @property
    def rows(self):
        return int(self._L.szg_index_rows(self._h))

    @property
    def live_rows(self):
        return int(self._L.szg_index_live_rows(self._h))

    # -- search -------------------------------------------------------------
    def _allow_arg(self, allow, n_queries):
        if allow is None:
            return None, None
        a = np.asarray(allow)
        if a.dtype != np.uint64:
            a = pack_allow_bits(a)
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(n_queries, -1)
        words = (self.rows + 63) // 64
        if a.shape[1] != words:
            raise ValueError("allow mask has %d words per query, index needs %d"
                             % (a.shape[1], words))
        return a, _u64(a)

    def search_topk(self, queries, k, allow=None):
        """Returns (rows uint64[nq,k], dist float64[nq,k], count int32[nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.shape[1] != self.dim:
            raise ValueError("query length %d != dimension %d" % (q.shape[1], self.dim))
        nq = q.shape[0]
        k = int(k)
        out_rows = np.zeros((nq, max(k, 0)), dtype=np.uint64)
        out_dist = np.zeros((nq, max(k, 0)), dtype=np.float64)
        out_count = np.zeros(nq, dtype=np.int32)
        keep, allow_p = self._allow_arg(allow, nq)
        check(self._L.szg_search_topk(self._h, _f64(q), nq, k, allow_p, _u64(out_rows),
                                      _f64(out_dist),
                                      out_count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
              "szg_search_topk")
        del keep
        return out_rows, out_dist, out_count

    def search_radius(self, query, radius, allow=None, capacity=None):
        """Returns (rows uint64[n], dist float64[n]); grows the buffer on SZG_E_TRUNCATED."""
        q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
        if q.size != self.dim:
            raise ValueError("query length %d != dimension %d" % (q.size, self.dim))
        keep, allow_p = self._allow_arg(allow, 1)
        cap = int(capacity) if capacity is not None else 1 << 16  # a too-small buffer costs a second sweep
        while True:
            out_rows = np.zeros(max(cap, 1), dtype=np.uint64)
            out_dist = np.zeros(max(cap, 1), dtype=np.float64)
            total = ctypes.c_uint64(0)
            rc = self._L.szg_search_radius(self._h, _f64(q), float(radius), allow_p,
                                           _u64(out_rows), _f64(out_dist), cap,
                                           ctypes.byref(total))
            if rc == _lib.SZG_E_TRUNCATED and capacity is None:
                cap = int(total.value)
                continue
            if rc == _lib.SZG_E_TRUNCATED:
                n = min(int(total.value), cap)
                return out_rows[:n], out_dist[:n], int(total.value)
            check(rc, "szg_search_radius")
            n = int(total.value)
            if capacity is None:
                return out_rows[:n], out_dist[:n]
            return out_rows[:n], out_dist[:n], n

    def _radius_csr(self, fn, name, queries, radii, allow, refetch=None):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.shape[1] != self.dim:
            raise ValueError("query length %d != dimension %d" % (q.shape[1], self.dim))
        nq = q.shape[0]
        rad = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float64), (nq,)))
        keep, allow_p = self._allow_arg(allow, nq)
        cap = max(1 << 16, nq << 12)  # (a truncated call is answered again from scratch: start generous)
        while True:
            out_rows = np.empty(max(cap, 1), dtype=np.uint64)
            out_dist = np.empty(max(cap, 1), dtype=np.float64)
            off = np.zeros(nq + 1, dtype=np.uint64)
            rc = fn(self._h, _f64(q), nq, _f64(rad), allow_p, _u64(out_rows), _f64(out_dist), cap, _u64(off))
            if rc == _lib.SZG_E_TRUNCATED:
                cap = int(off[nq])
                if refetch is None:
                    continue
                # a sharded call: the merged answer is kept by the communicator -- fetching it again is local (the
                # collective itself must never be repeated by some ranks only)
                out_rows = np.empty(max(cap, 1), dtype=np.uint64)
                out_dist = np.empty(max(cap, 1), dtype=np.float64)
                rc = refetch(nq, _u64(out_rows), _f64(out_dist), cap, _u64(off))
            check(rc, name)
            del keep
            return [(out_rows[int(off[i]):int(off[i + 1])], out_dist[int(off[i]):int(off[i + 1])]) for i in range(nq)]

    def search_radius_batch(self, queries, radii, allow=None):
        """Radius searches for a batch (radii: one value or one per query): a list of (rows, dist) per query,
        ascending distance.  The collect sweeps of the batch share query-major launches."""
        return self._radius_csr(self._L.szg_search_radius_batch, "szg_search_radius_batch", queries, radii, allow)

    # -- one process per GPU: the exchange inside the library (syzgydb_amd/sharded.py: Comm) ------------
    def attach_comm(self, comm):
        """Sharded searches of this handle go through `comm` (a sharded.Comm; None detaches)."""
        check(self._L.szg_index_attach_comm(self._h, comm._h if comm is not None else None), "szg_index_attach_comm")
        self._comm = comm  # keeps it alive as long as the handle uses it

    def search_topk_sharded(self, queries, k, allow=None):
        """Collective: the single-collection top-k over every rank's rows.
        Returns (rows uint64[nq,k] GLOBAL, dist float64[nq,k], count int32[nq], history_dependent bool[nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.shape[1] != self.dim:
            raise ValueError("query length %d != dimension %d" % (q.shape[1], self.dim))
        nq, k = q.shape[0], int(k)
        out_rows = np.zeros((nq, max(k, 0)), dtype=np.uint64)
        out_dist = np.zeros((nq, max(k, 0)), dtype=np.float64)
        out_count = np.zeros(nq, dtype=np.int32)
        hist = np.zeros(nq, dtype=np.uint8)
        keep, allow_p = self._allow_arg(allow, nq)
        check(self._L.szg_search_topk_sharded(self._h, _f64(q), nq, k, allow_p, _u64(out_rows), _f64(out_dist),
                                              out_count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _u8(hist)),
              "szg_search_topk_sharded")
        del keep
        return out_rows, out_dist, out_count, hist.astype(bool)

    def search_radius_sharded(self, queries, radii, allow=None):
        """Collective: radius searches over every rank's rows; a list of (rows GLOBAL, dist) per query."""
        comm = self._comm
        return self._radius_csr(self._L.szg_search_radius_sharded, "szg_search_radius_sharded", queries, radii, allow,
                                refetch=lambda nq, r, d, cap, off: self._L.szg_comm_last_radius(comm._h, nq, r, d, cap, off))

    # -- diagnostics ----------------------------------------------------------
    def set_timing(self, enabled):
        """False / 0 off; True / 1 events around the scan launches; 2 also around each batch's pipeline."""
        check(self._L.szg_set_timing(self._h, int(enabled)), "szg_set_timing")

    def stats(self):
        s = SzgStats()
        check(self._L.szg_get_stats(self._h, ctypes.byref(s)), "szg_get_stats")
        return {name: getattr(s, name) for name, _ in SzgStats._fields_}

    def reset_stats(self):
        check(self._L.szg_reset_stats(self._h), "szg_reset_stats")

    def set_option(self, name, value):
        check(self._L.szg_set_option(self._h, name.encode(), int(value)), "szg_set_option")
        self.options[name] = int(value)   # (what the host mirrors consult: e.g. tie_mode, collection.py)


def f64_probe(op, a, b=None):
    """Device float64 primitive probe (tests): 0 div, 1 sqrt, 2 Go acos, 3 round, 4 f32 narrow."""
    L = _lib.load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b if b is not None else a, dtype=np.float64)
    out = np.zeros_like(a)
    check(L.szg_debug_f64_probe(int(op), _f64(a), _f64(bb), _f64(out), a.size), "szg_debug_f64_probe")
    return out
