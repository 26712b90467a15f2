"""ctypes wrapper over oracle/libsyzgy_oracle.so (the CPU parity oracle).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from syzgydb_amd/.  See syzgy_oracle.h for
the parity-pinning status and the reference lines each function restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsyzgy_oracle.so")

EUCLIDEAN = 0  # collection.go:186-189
COSINE = 1


def build(force=False):
    """Compile the oracle with gcc (no reference sources involved)."""
    src = os.path.join(_HERE, "syzgy_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "libsyzgy_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        f64p = ctypes.POINTER(ctypes.c_double)
        L.orc_quantize.restype = ctypes.c_uint64
        L.orc_quantize.argtypes = [ctypes.c_double, ctypes.c_int]
        L.orc_dequantize.restype = ctypes.c_double
        L.orc_dequantize.argtypes = [ctypes.c_uint64, ctypes.c_int]
        L.orc_vector_size.restype = ctypes.c_int64
        L.orc_vector_size.argtypes = [ctypes.c_int, ctypes.c_int]
        L.orc_encode_vector.restype = None
        L.orc_encode_vector.argtypes = [f64p, ctypes.c_int, ctypes.c_int, u8p]
        L.orc_decode_vector.restype = None
        L.orc_decode_vector.argtypes = [u8p, ctypes.c_int, ctypes.c_int, f64p]
        L.orc_euclidean.restype = ctypes.c_double
        L.orc_euclidean.argtypes = [f64p, f64p, ctypes.c_int]
        L.orc_angular.restype = ctypes.c_double
        L.orc_angular.argtypes = [f64p, f64p, ctypes.c_int]
        L.orc_go_acos.restype = ctypes.c_double
        L.orc_go_acos.argtypes = [ctypes.c_double]
        L.orc_search_exact.restype = ctypes.c_int64
        L.orc_search_exact.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, f64p, ctypes.c_int, ctypes.c_double,
                                       u8p, u64p, f64p, ctypes.c_uint64, u64p]
        L.orc_all_distances.restype = None
        L.orc_all_distances.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, f64p, f64p]
        L.orc_distances_for_rows.restype = None
        L.orc_distances_for_rows.argtypes = [u8p, u64p, ctypes.c_uint64, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, f64p, f64p]
        L.orc_sorted_id_order.restype = None
        L.orc_sorted_id_order.argtypes = [u64p, ctypes.c_uint64, u64p]
        L.orc_splitmix64.restype = ctypes.c_uint64
        L.orc_splitmix64.argtypes = [ctypes.c_uint64]
        L.orc_synth_value.restype = ctypes.c_double
        L.orc_synth_value.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.orc_synth_vectors.restype = None
        L.orc_synth_vectors.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                        ctypes.c_int, f64p]
        L.orc_synth_rows.restype = None
        L.orc_synth_rows.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_int, ctypes.c_int, u8p]
        L.orc_bench_topk.restype = ctypes.c_double
        L.orc_bench_topk.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, f64p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, u64p]
        L.orc_spans_build.restype = ctypes.c_uint64
        L.orc_spans_build.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      u8p, ctypes.c_uint64, u64p]
        L.orc_bench_topk_faithful.restype = ctypes.c_double
        L.orc_bench_topk_faithful.argtypes = [u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, f64p, ctypes.c_int, ctypes.c_int, u64p]
        vp = ctypes.c_void_p
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.orc_lsh_build.restype = vp
        L.orc_lsh_build.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_uint64]
        L.orc_lsh_free.restype = None
        L.orc_lsh_free.argtypes = [vp]
        L.orc_lsh_sizes.restype = None
        L.orc_lsh_sizes.argtypes = [vp, i64p, i64p]
        L.orc_lsh_export.restype = None
        L.orc_lsh_export.argtypes = [vp, i32p, i32p, i32p, f64p, f64p, i64p, i32p, u64p]
        L.orc_lsh_search.restype = ctypes.c_int64
        L.orc_lsh_search.argtypes = [vp, ctypes.c_uint64, f64p, ctypes.c_int, ctypes.c_double, u8p, u64p, f64p,
                                     ctypes.c_uint64, u64p, u64p]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def quantize(value, bits):
    return int(lib().orc_quantize(float(value), int(bits)))


def dequantize(value, bits):
    return float(lib().orc_dequantize(int(value), int(bits)))


def vector_size(bits, dim):
    return int(lib().orc_vector_size(int(bits), int(dim)))


def encode_vector(vec, bits):
    v = _f64(vec)
    out = np.zeros(max(vector_size(bits, v.size), 0), dtype=np.uint8)
    lib().orc_encode_vector(_p(v, ctypes.c_double), v.size, bits, _p(out, ctypes.c_uint8))
    return out


def encode_rows(vectors, bits):
    V = _f64(vectors)
    n, dim = V.shape
    rb = vector_size(bits, dim)
    out = np.zeros((n, rb), dtype=np.uint8)
    for i in range(n):
        lib().orc_encode_vector(_p(V[i], ctypes.c_double), dim, bits,
                                out[i].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def decode_vector(data, dim, bits):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(dim, dtype=np.float64)
    lib().orc_decode_vector(_p(d, ctypes.c_uint8), dim, bits, _p(out, ctypes.c_double))
    return out


def euclidean(a, b):
    a, b = _f64(a), _f64(b)
    return float(lib().orc_euclidean(_p(a, ctypes.c_double), _p(b, ctypes.c_double), a.size))


def angular(a, b):
    a, b = _f64(a), _f64(b)
    return float(lib().orc_angular(_p(a, ctypes.c_double), _p(b, ctypes.c_double), a.size))


def go_acos(x):
    return float(lib().orc_go_acos(float(x)))


def search_exact(rows, dim, bits, metric, query, k=0, radius=0.0, allow=None, capacity=None):
    """Reference Search{Precision:"exact"} over packed rows in visit order.

    Returns (row_indices uint64[], distances float64[], points_searched).
    """
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    rb = vector_size(bits, dim)
    n_rows = rows.size // rb if rb > 0 else 0
    q = _f64(query)
    if capacity is None:
        capacity = n_rows if radius > 0 else min(max(k, 0), n_rows)
    capacity = max(int(capacity), 1)
    out_rows = np.zeros(capacity, dtype=np.uint64)
    out_dist = np.zeros(capacity, dtype=np.float64)
    searched = ctypes.c_uint64(0)
    allow_p = None
    if allow is not None:
        allow = np.ascontiguousarray(allow, dtype=np.uint8)
        assert allow.size == n_rows
        allow_p = _p(allow, ctypes.c_uint8)
    n = lib().orc_search_exact(_p(rows, ctypes.c_uint8) if rows.size else None, n_rows, dim,
                               bits, metric, _p(q, ctypes.c_double), int(k), float(radius),
                               allow_p, _p(out_rows, ctypes.c_uint64),
                               _p(out_dist, ctypes.c_double), capacity, ctypes.byref(searched))
    if n < 0:
        raise ValueError("orc_search_exact: bad arguments")
    m = min(int(n), capacity)
    return out_rows[:m].copy(), out_dist[:m].copy(), int(searched.value)


def all_distances(rows, dim, bits, metric, query):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    rb = vector_size(bits, dim)
    n_rows = rows.size // rb
    q = _f64(query)
    out = np.zeros(n_rows, dtype=np.float64)
    lib().orc_all_distances(_p(rows, ctypes.c_uint8), n_rows, dim, bits, metric,
                            _p(q, ctypes.c_double), _p(out, ctypes.c_double))
    return out


class GoHeap:
    """The reference's result heap, element for element (small cases only: pure Python).

    Go's container/heap (Push = append + up, Pop = swap(0, n) + down(0, n) + remove last) over
    resultPriorityQueue (collection.go:536-564): Less(i, j) = priority[i] > priority[j], a max-heap on distance.
    `items` is the heap's ARRAY -- (row, distance) pairs in container/heap's own order -- so that a replay can be
    continued from a state another party left (the rank-to-rank chain of csrc/scan_comm.cpp)."""

    def __init__(self, items=()):
        self.a = [(int(r), float(d)) for r, d in items]

    def _less(self, i, j):
        return self.a[i][1] > self.a[j][1]

    def _up(self, j):
        while True:
            i = (j - 1) // 2 if j > 0 else 0      # Go: (j - 1) / 2 truncates toward zero, so the parent of 0 is 0
            if i == j or not self._less(j, i):
                break
            self.a[i], self.a[j] = self.a[j], self.a[i]
            j = i

    def _down(self, i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            if j1 + 1 < n and self._less(j1 + 1, j1):
                j = j1 + 1
            if not self._less(j, i):
                break
            self.a[i], self.a[j] = self.a[j], self.a[i]
            i = j

    def push(self, row, dist):
        self.a.append((int(row), float(dist)))
        self._up(len(self.a) - 1)

    def pop(self):
        n = len(self.a) - 1
        self.a[0], self.a[n] = self.a[n], self.a[0]
        self._down(0, n)
        return self.a.pop()

    def consider_topk(self, row, dist, k):
        """consider()'s top-k branch for one visited record (collection.go:606-619)."""
        if len(self.a) <= k:
            if len(self.a) < k or self.a[0][1] > dist:
                self.push(row, dist)
                if len(self.a) > k:
                    self.pop()

    def items(self):
        return list(self.a)


def sorted_id_order(ids):
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    perm = np.zeros(ids.size, dtype=np.uint64)
    lib().orc_sorted_id_order(_p(ids, ctypes.c_uint64), ids.size, _p(perm, ctypes.c_uint64))
    return perm


def splitmix64(x):
    return int(lib().orc_splitmix64(int(x) & 0xFFFFFFFFFFFFFFFF))


def synth_vectors(seed, first_row, n_rows, dim):
    out = np.zeros((n_rows, dim), dtype=np.float64)
    lib().orc_synth_vectors(int(seed), int(first_row), int(n_rows), int(dim),
                            _p(out, ctypes.c_double))
    return out


def synth_rows(seed, first_row, n_rows, dim, bits):
    rb = vector_size(bits, dim)
    out = np.zeros((n_rows, rb), dtype=np.uint8)
    lib().orc_synth_rows(int(seed), int(first_row), int(n_rows), int(dim), int(bits),
                         _p(out, ctypes.c_uint8))
    return out


def bench_topk(rows, dim, bits, metric, queries, k, threads):
    """Time n_queries exact top-k searches; returns (seconds, rows[n_queries,k])."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    rb = vector_size(bits, dim)
    n_rows = rows.size // rb
    Q = _f64(queries).reshape(-1, dim)
    out = np.zeros((Q.shape[0], k), dtype=np.uint64)
    secs = lib().orc_bench_topk(_p(rows, ctypes.c_uint8), n_rows, dim, bits, metric,
                                _p(Q, ctypes.c_double), Q.shape[0], k, int(threads),
                                _p(out, ctypes.c_uint64))
    return float(secs), out


def bench_topk_faithful(rows, dim, bits, metric, queries, k, meta_len=16):
    """Single-thread exact top-k with the reference's per-record overheads (CRC32 of the
    span, span parse, fresh decode buffer); returns (seconds, rows[n_queries,k])."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    rb = vector_size(bits, dim)
    n_rows = rows.size // rb
    cap = n_rows * (rb + meta_len + 64)
    spans = np.zeros(cap, dtype=np.uint8)
    offsets = np.zeros(n_rows, dtype=np.uint64)
    used = lib().orc_spans_build(_p(rows, ctypes.c_uint8), n_rows, dim, bits, meta_len,
                                 _p(spans, ctypes.c_uint8), cap, _p(offsets, ctypes.c_uint64))
    assert used > 0
    Q = _f64(queries).reshape(-1, dim)
    out = np.zeros((Q.shape[0], k), dtype=np.uint64)
    secs = lib().orc_bench_topk_faithful(_p(spans, ctypes.c_uint8), _p(offsets, ctypes.c_uint64), n_rows,
                                         dim, bits, metric, _p(Q, ctypes.c_double), Q.shape[0], k,
                                         _p(out, ctypes.c_uint64))
    if secs < 0:
        raise RuntimeError("faithful baseline: span failed to parse")
    return float(secs), out


class LshForest:
    """The reference's LSH forest (lshtree.go:102-251 insert / split rules) over packed rows,
    built from a documented splitmix64 stream in place of Go's math/rand."""

    def __init__(self, rows, dim, bits, metric, threshold=100, num_trees=5, seed=1):
        self._rows = np.ascontiguousarray(rows, dtype=np.uint8)   # borrowed by the C side
        self.n_rows = self._rows.size // vector_size(bits, dim)
        self.dim, self.bits, self.metric, self.num_trees = dim, bits, metric, num_trees
        self._h = lib().orc_lsh_build(_p(self._rows, ctypes.c_uint8), self.n_rows, dim, bits, metric, threshold,
                                      num_trees, seed)
        if not self._h:
            raise ValueError("orc_lsh_build failed")

    def __del__(self):
        try:
            if self._h:
                lib().orc_lsh_free(self._h)
                self._h = None
        except Exception:
            pass

    def export(self):
        """Flat arrays: dict(roots, left, right, normals, b, ids_off, ids_cnt, ids)."""
        nn, ni = ctypes.c_int64(0), ctypes.c_int64(0)
        lib().orc_lsh_sizes(self._h, ctypes.byref(nn), ctypes.byref(ni))
        nn, ni = nn.value, ni.value
        out = dict(roots=np.zeros(self.num_trees, np.int32), left=np.zeros(nn, np.int32), right=np.zeros(nn, np.int32),
                   normals=np.zeros((nn, self.dim)), b=np.zeros(nn), ids_off=np.zeros(nn, np.int64),
                   ids_cnt=np.zeros(nn, np.int32), ids=np.zeros(max(ni, 1), np.uint64))
        lib().orc_lsh_export(self._h, _p(out["roots"], ctypes.c_int32), _p(out["left"], ctypes.c_int32),
                             _p(out["right"], ctypes.c_int32), _p(out["normals"], ctypes.c_double),
                             _p(out["b"], ctypes.c_double), _p(out["ids_off"], ctypes.c_int64),
                             _p(out["ids_cnt"], ctypes.c_int32), _p(out["ids"], ctypes.c_uint64))
        out["ids"] = out["ids"][:ni]
        return out

    def search(self, query, k=0, radius=0.0, allow=None):
        """lshTree.search + consider(): (rows, dist, points_searched, visit_order)."""
        q = _f64(query).reshape(-1)
        cap = self.n_rows
        out_rows = np.zeros(max(cap, 1), np.uint64)
        out_dist = np.zeros(max(cap, 1), np.float64)
        order = np.zeros(max(cap, 1), np.uint64)
        searched = ctypes.c_uint64(0)
        al = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint8)
        n = lib().orc_lsh_search(self._h, self.n_rows, _p(q, ctypes.c_double), int(k), float(radius),
                                 _p(al, ctypes.c_uint8) if al is not None else None,
                                 _p(out_rows, ctypes.c_uint64), _p(out_dist, ctypes.c_double), cap,
                                 ctypes.byref(searched), _p(order, ctypes.c_uint64))
        return out_rows[:n], out_dist[:n], int(searched.value), order[:int(searched.value)]
