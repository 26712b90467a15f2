"""Batch / query-major / coalesced radius search (szg_search_radius_batch, collection.go:598-605 per query)
against the oracle: ids, order (ties included) and float64 distances identical."""
import threading

import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import ScanIndex, SZG_COSINE, SZG_EUCLIDEAN
from syzgydb_amd import _lib

pytestmark = pytest.mark.gpu

SEED = 0x53595A4700000900


def check(hits, rows, dim, bits, metric, queries, radii, masks=None, dead=()):
    for i, (r, d) in enumerate(hits):
        allow = None if masks is None else masks[i].copy()
        if len(dead):
            allow = np.ones(rows.shape[0], bool) if allow is None else allow
            allow[list(dead)] = False
        er, ed, _ = orc.search_exact(rows, dim, bits, metric, queries[i], radius=float(radii[i]), allow=allow)
        assert [int(x) for x in r] == [int(x) for x in er], (i, len(r), len(er))
        assert (np.asarray(d) == np.asarray(ed)).all(), i


@pytest.mark.parametrize("bits,metric,dim,n", [(4, SZG_COSINE, 384, 6000), (8, SZG_EUCLIDEAN, 48, 5000),
                                               (16, SZG_COSINE, 24, 3000), (32, SZG_COSINE, 96, 7000),
                                               (32, SZG_EUCLIDEAN, 768, 2500), (64, SZG_EUCLIDEAN, 17, 1500)])
def test_batch_matches_oracle_per_query_radii(bits, metric, dim, n):
    rows = orc.synth_rows(SEED + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 1, 0, 37, dim)   # 37: two full launches of 16 sweeps and a ragged third
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        # radii around each query's 5th..200th neighbour so that hit counts differ widely (incl. zero hits)
        _, dd, _ = ix.search_topk(Q, 200)
        radii = np.array([dd[i, [4, 30, 199, 0][i % 4]] * (0.5 if i % 11 == 10 else 1.0) for i in range(len(Q))])
        radii = np.maximum(radii, 1e-9)
        hits = ix.search_radius_batch(Q, radii)
        check(hits, rows, dim, bits, metric, Q, radii)
        assert sum(len(r) for r, _ in hits) > 500
        # the same through the single-query entry point (a batch of one each)
        for i in (0, 5, 36):
            r, d = ix.search_radius(Q[i], radii[i])
            assert (r == hits[i][0]).all() and (d == hits[i][1]).all()
        st = ix.stats()
        assert st["queries"] >= 37 + 3


def test_masks_tombstones_and_two_shards():
    dim, bits, metric, n = 64, 8, SZG_COSINE, 9000
    rows = orc.synth_rows(SEED + 50, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 51, 0, 20, dim)
    rng = np.random.default_rng(5)
    masks = rng.random((len(Q), n)) < np.linspace(0.02, 0.95, len(Q))[:, None]   # selective ... mild
    dead = [3, 64, 65, 4000, n - 1]
    for devices in ([0], [0, 0]):
        with ScanIndex(dim, bits, metric, devices=devices) as ix:
            ix.load(rows)
            for r in dead:
                ix.tombstone(r)
            radii = np.full(len(Q), 0.47)
            hits = ix.search_radius_batch(Q, radii, allow=masks)
            check(hits, rows, dim, bits, metric, Q, radii, masks=masks, dead=dead)
            hits = ix.search_radius_batch(Q, radii)          # tombstones only
            check(hits, rows, dim, bits, metric, Q, radii, dead=dead)
            for mask_dense in (0, 1):
                ix.set_option("mask_dense", mask_dense)
                hits = ix.search_radius_batch(Q[:5], radii[:5], allow=masks[:5])
                check(hits, rows, dim, bits, metric, Q[:5], radii[:5], masks=masks[:5], dead=dead)


def test_more_hits_than_the_batch_buffers_hold_and_ties():
    # a duplicate-heavy 4-bit corpus: thousands of hits per query, many equal distances; the first batch's buffers
    # (1024 entries per sweep) overflow, those queries are swept again on their own, later batches have room
    dim, bits, metric, n = 2, 4, SZG_EUCLIDEAN, 20000
    rows = orc.synth_rows(SEED + 60, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 61, 0, 40, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        radii = np.where(np.arange(len(Q)) % 3 == 0, 0.9, 0.05)
        for _ in range(2):
            hits = ix.search_radius_batch(Q, radii)
            check(hits, rows, dim, bits, metric, Q, radii)
        assert max(len(r) for r, _ in hits) > 4096


def test_truncated_capacity_reports_offsets():
    import ctypes
    dim, bits, metric, n = 32, 32, SZG_COSINE, 4000
    rows = orc.synth_rows(SEED + 70, 0, n, dim, bits)
    Q = np.ascontiguousarray(orc.synth_vectors(SEED + 71, 0, 6, dim))
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        radii = np.full(6, 0.46)
        full = ix.search_radius_batch(Q, radii)
        total = sum(len(r) for r, _ in full)
        assert total > 20
        cap = total // 2
        out_rows = np.zeros(cap, np.uint64)
        out_dist = np.zeros(cap, np.float64)
        off = np.zeros(7, np.uint64)
        u64, f64 = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)
        rc = ix._L.szg_search_radius_batch(ix._h, Q.ctypes.data_as(f64), 6, radii.ctypes.data_as(f64), None,
                                           out_rows.ctypes.data_as(u64), out_dist.ctypes.data_as(f64), cap,
                                           off.ctypes.data_as(u64))
        assert rc == _lib.SZG_E_TRUNCATED
        assert int(off[6]) == total
        flat = np.concatenate([r for r, _ in full])
        assert (out_rows == flat[:cap]).all()
        # radius <= 0 is listing / top-k territory (collection.go:598): refused
        bad = np.array([0.4, 0.0, 0.4, 0.4, 0.4, 0.4])
        rc = ix._L.szg_search_radius_batch(ix._h, Q.ctypes.data_as(f64), 6, bad.ctypes.data_as(f64), None,
                                           out_rows.ctypes.data_as(u64), out_dist.ctypes.data_as(f64), cap,
                                           off.ctypes.data_as(u64))
        assert rc == _lib.SZG_E_INVALID


def test_concurrent_radius_callers_are_coalesced():
    """The reference's Searches run concurrently under RLock (collection.go:570): radius callers with one
    query each -- and top-k callers beside them -- share query-major launches; every caller gets its own answer."""
    dim, bits, metric, n = 384, 4, SZG_COSINE, 40000
    rows = orc.synth_rows(SEED + 80, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 81, 0, 48, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        _, dd, _ = ix.search_topk(Q, 60)
        radii = dd[:, 59]
        out = [None] * len(Q)
        errs = []
        gate = threading.Barrier(len(Q))   # all callers at once (threads started one after the other may never meet)

        def work(i):
            try:
                gate.wait()
                if i % 6 == 5:
                    out[i] = ix.search_topk(Q[i], 7)
                else:
                    out[i] = ix.search_radius(Q[i], radii[i])
            except Exception as e:  # pragma: no cover
                errs.append(e)
        ix.reset_stats()
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(Q))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(len(Q)):
            if i % 6 == 5:
                er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=7)
                r, d, c = out[i]
                assert [int(x) for x in r[0, :c[0]]] == [int(x) for x in er] and (d[0, :c[0]] == ed).all()
            else:
                er, ed, _ = orc.search_exact(rows, dim, bits, metric, Q[i], radius=float(radii[i]))
                assert [int(x) for x in out[i][0]] == [int(x) for x in er] and (out[i][1] == ed).all()
        st = ix.stats()
        n_radius = sum(1 for i in range(len(Q)) if i % 6 != 5)
        assert st["scan_launches"] < n_radius + 8, st   # fewer collect launches than radius callers: sweeps were shared


def test_empty_collection_and_empty_batch():
    with ScanIndex(8, 32, SZG_COSINE) as ix:
        hits = ix.search_radius_batch(np.zeros((3, 8)) + 0.5, 0.4)   # empty collection: no results
        assert len(hits) == 3 and all(len(r) == 0 and len(d) == 0 for r, d in hits)
        ix.load(orc.synth_rows(SEED, 0, 10, 8, 32))
        assert ix.search_radius_batch(np.zeros((0, 8)), 0.4) == []


@pytest.mark.parametrize("bits,metric,dim,n", [(4, SZG_COSINE, 384, 6000), (8, SZG_COSINE, 768, 2500), (8, SZG_EUCLIDEAN, 100, 5000),
                                               (16, SZG_COSINE, 64, 4000), (16, SZG_EUCLIDEAN, 36, 3000),
                                               (32, SZG_COSINE, 96, 7000), (32, SZG_EUCLIDEAN, 33, 4000),
                                               (32, SZG_COSINE, 2, 5000),
                                               (64, SZG_COSINE, 48, 4000), (64, SZG_EUCLIDEAN, 17, 3000)])
def test_radius_batches_share_one_sweep(bits, metric, dim, n):
    """A radius batch of 2+ queries takes ONE shared sweep (the radius is the collect threshold, widened by the
    sweep's own error bound); answers identical to the oracle's and to the one-sweep-per-query form's -- with
    filters, radii that admit nothing, everything, or more hits than the batch's buffers hold."""
    rows = orc.synth_rows(SEED + 300 + bits + dim, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 301 + dim, 0, 110, dim)   # 110: a full batch of 96 and a ragged one
    if dim == 2:  # rows and queries within a few degrees of one direction (the bfloat16 band's worst case)
        rng0 = np.random.default_rng(9)
        rows = orc.encode_rows(rng0.uniform(-1, 1, (n, dim)) * 1e3 + 5e3, bits)
        Q = rng0.uniform(-1, 1, (110, dim)) * 1e3 + 5e3
    rng = np.random.default_rng(bits + dim)
    masks = rng.random((len(Q), n)) < 0.7
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        _, dd, _ = ix.search_topk(Q, 150)
        radii = np.array([dd[i, [4, 30, 149, 0][i % 4]] for i in range(len(Q))])
        radii = np.maximum(radii, 1e-12)
        radii[7] *= 0.25                                       # (usually) nothing
        radii[11] = 1.0 if metric == SZG_COSINE else 1e30      # everything: overflows the batch's buffers
        radii[12] = float(dd[12, -1]) * 1.5                    # many
        ix.reset_stats()
        hits = ix.search_radius_batch(Q, radii)
        st = ix.stats()
        check(hits, rows, dim, bits, metric, Q, radii)
        assert len(hits[11][0]) == n
        if not __import__("os").environ.get("SZG_OPTIONS"):
            assert st["mq_queries"] == 110 and st["mq_launches"] >= 2
        hm = ix.search_radius_batch(Q[:40], radii[:40], allow=masks[:40])
        check(hm, rows, dim, bits, metric, Q[:40], radii[:40], masks=masks[:40])
        ix.set_option("radius_mq", 0)
        ix.reset_stats()
        h0 = ix.search_radius_batch(Q, radii)
        assert ix.stats()["mq_queries"] == 0
        for a, b in zip(hits, h0):
            assert (a[0] == b[0]).all() and (np.asarray(a[1]) == np.asarray(b[1])).all()


@pytest.mark.parametrize("dim", [1, 3, 9, 36])
@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
def test_radius_batch_16bit_rows_with_padding_codes_and_small_queries(dim, metric):
    """16-bit rows whose last 16-byte piece holds padding codes (they decode to -65535), queries and neighbours of
    small norm: the shared sweep's row norms must not carry the padding (fuzz seeds 341 / 342: a norm formed as
    `total - 7 x 65535^2` loses every bit of a small row's norm, and radius hits with it)."""
    n = 3000
    rng = np.random.default_rng(100 + dim)
    vec = rng.uniform(-1, 1, (n, dim))
    Q = rng.uniform(-1, 1, (12, dim)) * np.array([1e-3, 1e-2, 0.1, 1.0] * 3)[:, None]
    for j in range(12):   # a few rows right next to each query
        vec[rng.integers(0, n, 6)] = Q[j] * (1.0 + rng.uniform(-0.3, 0.3, (6, 1)))
    rows = orc.encode_rows(vec, 16)
    radii = []
    for j in range(12):
        od = orc.search_exact(rows, dim, 16, metric, Q[j], k=[2, 5, 40][j % 3])[1]
        fin = [x for x in od if x == x and x > 0]
        radii.append(float(fin[-1]) if fin else 0.5)
    with ScanIndex(dim, 16, metric) as ix:
        ix.load(rows)
        hits = ix.search_radius_batch(Q, radii)
        check(hits, rows, dim, 16, metric, Q, radii)
        r, d, c = ix.search_topk(Q, 5)
        for j in range(12):
            er, ed, _ = orc.search_exact(rows, dim, 16, metric, Q[j], k=5)
            assert [int(x) for x in r[j, : c[j]]] == [int(x) for x in er] and (d[j, : c[j]] == np.asarray(ed)).all(), j


@pytest.mark.parametrize("bits,metric,dim,n,nth", [(4, SZG_EUCLIDEAN, 768, 5000, 60), (4, SZG_EUCLIDEAN, 384, 1000, 61),
                                                   (8, SZG_COSINE, 2, 3000, 60), (4, SZG_EUCLIDEAN, 100, 65, 56)])
def test_hits_whose_distances_differ_in_their_last_bits_keep_the_references_order(bits, metric, dim, n, nth):
    """Coarse rows (4-bit Euclidean, 2-dimensional 8-bit cosine): a result holds exactly equal distances AND distances
    that differ only in their last bits.  The device-side sort of a batch's hits compares 64-bit ordered keys; round 4
    briefly clamped them through min(), whose overload took uint64 through double -- 11 bits gone, near-equal
    distances out of order, nothing downstream looks at a list it believes sorted (scripts/fuzz_gpu.py seed 401: ten
    of 6 443 cases).  One sweep per query and shared sweeps alike."""
    rows = orc.synth_rows(SEED + 700 + dim, 0, n, dim, bits)
    Q = orc.synth_vectors(SEED + 701 + dim, 0, 17, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        radii = []
        for i in range(len(Q)):
            _, od, _ = orc.search_exact(rows, dim, bits, metric, Q[i], k=min(nth, n))
            fin = [x for x in od if x == x and x > 0]
            radii.append(float(fin[-1]) if fin else 0.5)
        for shared in (1, 0):
            ix.set_option("radius_mq", shared)
            check(ix.search_radius_batch(Q, radii), rows, dim, bits, metric, Q, radii)
            r1, d1 = ix.search_radius(Q[3], radii[3])
            check([(r1, d1)], rows, dim, bits, metric, Q[3:4], radii[3:4])
