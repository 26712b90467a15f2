"""Host-side logic that needs no GPU: the numpy codec, the query synthesiser, the
C-ABI library's exports, error paths and the cross-shard merge."""
import ctypes
import os
import re

import numpy as np
import pytest

from golden_util import load_kats, same_f64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_codec_matches_oracle(oracle):
    from syzgydb_amd import codec
    rng = np.random.default_rng(0)
    V = rng.uniform(-1.3, 1.3, (64, 9))
    V[0, :] = [0.0, 1.0, -1.0, 0.5, -0.5, 1e-9, -1e-9, 0.999999, -0.999999]
    for bits in (4, 8, 16, 32, 64):
        enc = codec.encode_rows(V, bits)
        assert (enc == oracle.encode_rows(V, bits)).all(), bits
        dec = codec.decode_rows(enc, 9, bits)
        ref = np.stack([oracle.decode_vector(enc[i], 9, bits) for i in range(V.shape[0])])
        assert same_f64(dec, ref), bits
        assert codec.vector_size(bits, 9) == oracle.vector_size(bits, 9)
    with pytest.raises(ValueError):
        codec.vector_size(7, 3)


def test_codec_kats():
    from syzgydb_amd import codec
    kat = load_kats()["encode_vector"]
    for bits, hx in kat["bytes_hex"].items():
        assert codec.encode_rows([kat["vector"]], int(bits)).tobytes().hex() == hx
    # half away from zero, incl. the value just below a tie
    assert int(codec.quantize(np.array([0.0]), 4)[0]) == 8
    below = np.nextafter(0.5, 0) * 2 / 15 - 1  # (v+1)/2*15 just below 0.5
    assert int(codec.quantize(np.array([below]), 4)[0]) == 0


def test_synth_matches_oracle(oracle):
    from syzgydb_amd.synth import splitmix64, synth_vectors
    for seed, first, n, dim in ((1, 0, 3, 4), (0x53595A4700000003, 7, 5, 11), (2**63 + 5, 1000, 2, 768)):
        assert (synth_vectors(seed, first, n, dim) == oracle.synth_vectors(seed, first, n, dim)).all()
    assert int(splitmix64(np.array([12345], dtype=np.uint64))[0]) == oracle.splitmix64(12345)


def _declared_symbols(header="syzgy_scan.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(szg_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from syzgydb_amd import _lib
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libsyzgy_scan.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared
    assert L.szg_abi_version() == 4
    pager = _declared_symbols("syzgy_pager.h")
    assert sorted(_lib.PAGER_EXPORTS) == pager
    for name in pager:
        assert hasattr(L, name), "libsyzgy_scan.so does not export %s" % name


def test_error_paths_without_gpu_work():
    from syzgydb_amd import _lib
    L = _lib.load()
    assert L.szg_strerror(0) == b"ok"
    assert b"capacity" in L.szg_strerror(_lib.SZG_E_TRUNCATED)
    assert L.szg_row_bytes(4, 3) == 2 and L.szg_row_bytes(32, 768) == 3072
    assert L.szg_row_bytes(7, 3) == -1 and L.szg_row_bytes(8, 0) == -1
    h = ctypes.c_void_p()
    # bad arguments are rejected before any device work
    assert L.szg_index_create(ctypes.byref(h), 0, 32, 1, None, 0) == _lib.SZG_E_INVALID
    assert L.szg_index_create(ctypes.byref(h), 8, 5, 1, None, 0) == _lib.SZG_E_INVALID
    assert L.szg_index_create(ctypes.byref(h), 8, 32, 2, None, 0) == _lib.SZG_E_INVALID
    assert b"distance method" in L.szg_last_error()
    assert L.szg_index_create(None, 8, 32, 1, None, 0) == _lib.SZG_E_INVALID


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from syzgydb_amd import ScanIndex, SzgError, _lib
    with pytest.raises(SzgError) as e:
        ScanIndex(8, 32, 1)
    assert e.value.code == _lib.SZG_E_NODEVICE


def test_merge_topk_matches_oracle_over_shards(oracle):
    from syzgydb_amd.sharded import merge_topk, shard_range
    dim, bits, n, k, G = 12, 32, 1000, 10, 3
    rows = oracle.synth_rows(5, 0, n, dim, bits)
    Q = oracle.synth_vectors(6, 0, 4, dim)
    for metric in (0, 1):
        R = np.zeros((G, Q.shape[0], k + 1), np.uint64)
        D = np.zeros((G, Q.shape[0], k + 1), np.float64)
        C = np.zeros((G, Q.shape[0]), np.int32)
        for g in range(G):
            lo, hi = shard_range(n, g, G)
            assert lo % 64 == 0
            for qi in range(Q.shape[0]):
                r, d, _ = oracle.search_exact(rows[lo:hi], dim, bits, metric, Q[qi], k=k + 1)
                R[g, qi, : len(r)] = r + lo
                D[g, qi, : len(r)] = d
                C[g, qi] = len(r)
        out_r, out_d, out_c, hist = merge_topk(k, R, D, C)
        for qi in range(Q.shape[0]):
            r, d, _ = oracle.search_exact(rows, dim, bits, metric, Q[qi], k=k)
            assert [int(x) for x in out_r[qi, : out_c[qi]]] == [int(x) for x in r]
            assert same_f64(out_d[qi, : out_c[qi]], d)
            assert not hist[qi]


def test_merge_topk_flags_ties_and_short_lists():
    from syzgydb_amd.sharded import merge_topk
    R = np.array([[[0, 1, 2]], [[64, 65, 66]]], np.uint64)
    D = np.array([[[1.0, 2.0, 9.0]], [[2.0, 3.0, 9.0]]])
    C = np.array([[2], [2]], np.int32)
    r, d, c, hist = merge_topk(2, R, D, C)
    assert list(r[0]) == [0, 1] and list(d[0]) == [1.0, 2.0] and hist[0]  # 2.0 == 2.0 at the boundary
    r, d, c, hist = merge_topk(5, R, D, C)
    assert c[0] == 4 and r[0, 4] == np.iinfo(np.uint64).max


def test_merge_topk_records_is_the_same_merge():
    """The records form (what the ranks all-gather: rows | distance bits | count per query)
    gives exactly szg_merge_topk's answer, NaN distances and short lists included."""
    from syzgydb_amd.sharded import merge_topk, merge_topk_records
    rng = np.random.default_rng(5)
    for case in range(40):
        G, nq, kk = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.integers(1, 12))
        k = max(1, kk - 1)
        R = rng.integers(0, 1 << 40, (G, nq, kk)).astype(np.uint64)
        D = np.round(rng.uniform(0, 2, (G, nq, kk)), 1 if case % 2 else 6)
        if case % 5 == 0:
            D[0, 0, 0] = np.nan
        D.sort(axis=2)
        C = rng.integers(0, kk + 1, (G, nq)).astype(np.int32)
        rec = np.zeros((G, nq, 2 * kk + 1), np.int64)
        rec[:, :, :kk] = R.view(np.int64)
        rec[:, :, kk:2 * kk] = D.view(np.int64)
        rec[:, :, 2 * kk] = C
        a = merge_topk(k, R, D, C)
        b = merge_topk_records(k, rec, kk)
        assert (a[0] == b[0]).all() and same_f64(a[1], b[1]) and (a[2] == b[2]).all() and (a[3] == b[3]).all()


def test_shard_ranges_cover_everything():
    from syzgydb_amd.sharded import shard_range
    for n in (0, 1, 63, 64, 65, 1000, 1_000_000, 10_000_001):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n


def test_merge_topk_randomized(oracle):
    """Random shard counts, sizes, k and corpora (incl. grids with many equal distances):
    whenever szg_merge_topk does not flag the query as history-dependent, the merged list is the
    single-collection answer; when it does, it is still a correct top-k by distance."""
    from syzgydb_amd.sharded import merge_topk, shard_range
    rng = np.random.default_rng(77)
    for case in range(60):
        dim = int(rng.choice([2, 5, 16]))
        bits = int(rng.choice([4, 8, 32]))
        metric = int(rng.integers(0, 2))
        n = int(rng.choice([3, 64, 65, 200, 1500]))
        G = int(rng.integers(1, 9))
        k = int(rng.choice([1, 3, 10, 40]))
        vec = rng.uniform(-1, 1, (n, dim))
        if case % 3 == 0:
            vec = np.round(vec * 2) / 2          # many ties
        rows = oracle.encode_rows(vec, bits)
        Q = rng.uniform(-1, 1, (3, dim))
        kk = k + 1
        R = np.full((G, 3, kk), np.iinfo(np.uint64).max, np.uint64)
        D = np.zeros((G, 3, kk))
        C = np.zeros((G, 3), np.int32)
        for g in range(G):
            lo, hi = shard_range(n, g, G)
            for qi in range(3):
                if hi > lo:
                    r, d, _ = oracle.search_exact(rows[lo:hi], dim, bits, metric, Q[qi], k=kk)
                    R[g, qi, : len(r)] = r + lo
                    D[g, qi, : len(r)] = d
                    C[g, qi] = len(r)
        out_r, out_d, out_c, hist = merge_topk(k, R, D, C)
        for qi in range(3):
            r, d, _ = oracle.search_exact(rows, dim, bits, metric, Q[qi], k=k)
            got_d = out_d[qi, : out_c[qi]]
            assert out_c[qi] == len(r), (case, qi)
            if not hist[qi]:
                assert [int(x) for x in out_r[qi, : out_c[qi]]] == [int(x) for x in r], (case, qi)
                assert same_f64(got_d, d)
            else:   # equal distances / NaN among the best k+1: the multiset of distances is pinned
                a, b = np.sort(got_d), np.sort(d)
                assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (case, qi)


def test_bench_compact_line_keeps_the_contract_fields_and_one_short_object_per_leg():
    """bench.py prints ONE compact JSON line (a driver may keep only a tail of stdout): the contract's fields in full,
    one short object per extra leg, the rest goes to stderr."""
    import importlib.util, json, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rf = {"bound": "hbm", "achieved": 7000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.875, "traffic": None,
          "kernel": "k", "bytes_per_launch": 1, "sweeps_per_launch": 16.0, "avg_launch_ms": 1.0, "launches": 3}
    obj = {
        "metric": "m", "value": 1.0, "unit": "queries/s", "roofline": dict(rf), "cpu_baseline": {"value": 1.0},
        "batched": {"value": 2.0, "unit": "queries/s", "queries_per_sweep": 96.0, "avg_sweep_ms": 0.46,
                    "kernel": "szg::k<6> (mfma)", "roofline": dict(rf), "ids_identical_to_single_query_path": True,
                    "end_to_end_hbm_frac": 0.6},
        "lone_call": {"what": "long text " * 20, "ms": 0.5, "queries_per_s": 2000.0, "hbm_frac": 0.77, "ms_min": 0.49,
                      "ms_p90": 0.52},
        "spread": {"min": 1.0, "max": 1.1, "repeats": 5, "what": "text"},
        "batched_quantized": {"8bit": {"value": 3.0, "avg_pass_ms": 0.12, "roofline": dict(rf), "kernel": "x",
                                       "ids_and_distances_identical_to_single_query_path": True},
                              "4bit": {"error": "RuntimeError: x"}},
        "other_workloads": {"cfg5": {"value": 4.0, "roofline": dict(rf),
                                     "parity": {"identical_to_oracle": 2, "queries_checked": 2},
                                     "batched": {"value": 50000.0}},
                            "cfg2": {"error": "boom"}},
        "host_us_breakdown": {"x": 1},
    }
    c = bench.compact(obj)
    line = json.dumps(c)
    assert json.loads(line)["roofline"] == rf                      # the contract's object untouched
    assert c["batched"]["kernel"] == "szg::k<6>" and c["batched"]["end_to_end_hbm_frac"] == 0.6
    assert c["lone_call"] == {"ms": 0.5, "queries_per_s": 2000.0, "hbm_frac": 0.77}
    assert c["spread"] == {"min": 1.0, "max": 1.1, "repeats": 5}
    assert c["batched_quantized"]["8bit"] == {"value": 3.0, "avg_pass_ms": 0.12, "queries_per_pass": None,
                                              "roofline": bench._rf(rf), "identical_to_single_query_path": True}
    assert c["batched_quantized"]["4bit"] == {"error": "RuntimeError: x"}
    assert c["other_workloads"]["cfg5"]["batched_96_queries_per_s"] == 50000.0
    assert c["other_workloads"]["cfg5"]["identical_to_oracle"] == "2/2" and c["other_workloads"]["cfg2"] == {"error": "boom"}
    assert "host_us_breakdown" not in c and len(line) < 4000
