"""GPU parity: the HIP scan (through the C ABI) against the CPU oracle.

IDs must be bit-exact and in the same order; distances are the reference's
own float64 values, so they are compared for exact equality (the contract's
tolerance, 1e-5 relative for float32 corpora, is the fallback the asserts
name when a platform's float64 sqrt/div differs in the last bit).
"""
import os

import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import ScanIndex, SZG_COSINE, SZG_EUCLIDEAN, f64_probe

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: distances within 1e-5 relative (float32)

SEED = 0x53595A4700000000


def assert_same(rows, dist, o_rows, o_dist):
    assert len(rows) == len(o_rows)
    assert list(map(int, rows)) == list(map(int, o_rows)), "doc rows differ"
    d = np.asarray(dist, dtype=np.float64)
    od = np.asarray(o_dist, dtype=np.float64)
    both_nan = np.isnan(d) & np.isnan(od)
    ok = both_nan | (np.abs(d - od) <= REL_TOL * np.abs(od))
    assert ok.all(), (d, od)
    # stronger, expected: bit-identical float64
    assert (both_nan | (d == od)).all(), ("not bit-exact", d, od)


@pytest.mark.parametrize("bits", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
@pytest.mark.parametrize("dim,n", [(3, 10), (3, 1000), (17, 777), (128, 5000), (384, 3000), (768, 2000)])
def test_topk_matches_oracle(bits, metric, dim, n):
    rows = orc.synth_rows(SEED + bits, 0, n, dim, bits)
    queries = orc.synth_vectors(SEED + 1, 0, 4, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        assert ix.rows == n
        for mq in (0, 1):   # one sweep per query; the shared sweep (batches of >= 2 take it by default)
            ix.set_option("multi_query", mq)
            for k in (1, 10):
                r, d, c = ix.search_topk(queries, k)
                for qi in range(queries.shape[0]):
                    o_rows, o_dist, searched = orc.search_exact(rows, dim, bits, metric, queries[qi], k=k)
                    assert searched == n
                    assert c[qi] == len(o_rows)
                    assert_same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist)


@pytest.mark.parametrize("bits", [4, 8, 32])
@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
def test_synth_matches_oracle_bytes(bits, metric):
    dim, n = 37, 513
    with ScanIndex(dim, bits, metric) as ix:
        ix.synth(n, SEED + 7, first_row=100)
        got = ix.read_rows(0, n)
    want = orc.synth_rows(SEED + 7, 100, n, dim, bits)
    assert (got == want).all()


@pytest.mark.parametrize("bits", [4, 8, 16, 32, 64])
def test_load_read_roundtrip(bits):
    dim, n = 21, 300
    rows = orc.synth_rows(SEED, 0, n, dim, bits)
    with ScanIndex(dim, bits, SZG_EUCLIDEAN) as ix:
        ix.load(rows)
        assert (ix.read_rows(0, n) == rows).all()
        assert (ix.read_rows(17, 5) == rows[17:22]).all()


def test_f64_primitives_bit_exact():
    rng = np.random.default_rng(1)
    a = rng.uniform(1e-3, 1e3, 20000)
    b = rng.uniform(1e-3, 1e3, 20000)
    assert (f64_probe(0, a, b) == a / b).all()
    assert (f64_probe(1, a) == np.sqrt(a)).all()
    x = np.concatenate([rng.uniform(-1, 1, 20000), [1.0, -1.0, 0.0, 1.0000000000000002, 0.7, 0.66]])
    got = f64_probe(2, x)
    want = np.array([orc.go_acos(v) for v in x])
    assert ((got == want) | (np.isnan(got) & np.isnan(want))).all()


@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
def test_radius_matches_oracle(metric):
    dim, n, bits = 16, 4000, 32
    rows = orc.synth_rows(SEED, 0, n, dim, bits)
    q = orc.synth_vectors(SEED + 1, 0, 1, dim)[0]
    alld = orc.all_distances(rows, dim, bits, metric, q)
    for frac in (0.001, 0.05, 0.5):
        radius = float(np.quantile(alld, frac))
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, q, radius=radius)
        with ScanIndex(dim, bits, metric) as ix:
            ix.load(rows)
            r, d = ix.search_radius(q, radius)
        assert_same(r, d, o_rows, o_dist)
        assert (d <= radius).all()


def test_filter_mask_and_tombstones():
    dim, n, bits = 8, 500, 8
    rows = orc.synth_rows(SEED, 0, n, dim, bits)
    q = orc.synth_vectors(SEED + 1, 0, 1, dim)[0]
    allow = (np.arange(n) % 2 == 0)
    with ScanIndex(dim, bits, SZG_COSINE) as ix:
        ix.load(rows)
        r, d, c = ix.search_topk(q, 5, allow=allow)
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, SZG_COSINE, q, k=5,
                                             allow=allow.astype(np.uint8))
        assert_same(r[0, : c[0]], d[0, : c[0]], o_rows, o_dist)
        assert all(int(x) % 2 == 0 for x in r[0, : c[0]])
        # tombstone the best hit: it must disappear
        best = int(r[0, 0])
        ix.tombstone(best)
        assert ix.live_rows == n - 1
        allow2 = allow.copy()
        allow2[best] = False
        r2, d2, c2 = ix.search_topk(q, 5, allow=allow)
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, SZG_COSINE, q, k=5,
                                             allow=allow2.astype(np.uint8))
        assert_same(r2[0, : c2[0]], d2[0, : c2[0]], o_rows, o_dist)


def test_k_larger_than_n_and_empty():
    dim, bits = 5, 64
    rows = orc.synth_rows(SEED, 0, 7, dim, bits)
    q = orc.synth_vectors(SEED + 1, 0, 1, dim)[0]
    with ScanIndex(dim, bits, SZG_EUCLIDEAN) as ix:
        r, d, c = ix.search_topk(q, 3)
        assert c[0] == 0
        ix.load(rows)
        r, d, c = ix.search_topk(q, 20)
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, SZG_EUCLIDEAN, q, k=20)
        assert c[0] == 7
        assert_same(r[0, :7], d[0, :7], o_rows, o_dist)


def test_escalation_path_is_exact():
    """Many duplicate rows force ties across the candidate boundary (rule: a rare
    data-dependent branch needs its own test); force_escalate drives the same path
    on ordinary data."""
    dim, bits = 6, 32
    base = orc.synth_vectors(SEED, 0, 4, dim)
    vecs = np.repeat(base, 300, axis=0)  # 1200 rows, 4 distinct vectors
    rows = orc.encode_rows(vecs, bits)
    q = base[2] + 0.01
    for metric in (SZG_EUCLIDEAN, SZG_COSINE):
        with ScanIndex(dim, bits, metric) as ix:
            ix.load(rows)
            r, d, c = ix.search_topk(q, 10)
            st = ix.stats()
            assert st["escalations"] >= 1
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, q, k=10)
            assert_same(r[0, : c[0]], d[0, : c[0]], o_rows, o_dist)
    rows = orc.synth_rows(SEED, 0, 3000, 24, 8)
    q = orc.synth_vectors(SEED + 1, 0, 1, 24)[0]
    with ScanIndex(24, 8, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("force_escalate", 1)
        r, d, c = ix.search_topk(q, 10)
        assert ix.stats()["escalations"] == 1
        o_rows, o_dist, _ = orc.search_exact(rows, 24, 8, SZG_COSINE, q, k=10)
        assert_same(r[0, : c[0]], d[0, : c[0]], o_rows, o_dist)


@pytest.mark.parametrize("bits", [4, 8, 16, 32, 64])
def test_device_side_quantize_and_pack(bits):
    """szg_index_append_f64 == encodeDocument on the host (collection.go:713-743)."""
    dim, n = 19, 400
    rng = np.random.default_rng(bits)
    V = rng.uniform(-1.4, 1.4, (n, dim))
    V[0, :5] = [0.0, 1.0, -1.0, 0.5, -0.5]
    V[1, :3] = [np.nextafter(0.5, 0) * 2 / 15 - 1, 1e-300, -1e-300]
    with ScanIndex(dim, bits, SZG_EUCLIDEAN) as ix:
        ix.append_vectors(V[:150])
        ix.append_vectors(V[150:])
        assert ix.rows == n
        assert (ix.read_rows(0, n) == orc.encode_rows(V, bits)).all()


@pytest.mark.parametrize("bits,metric", [(4, 1), (8, 0), (32, 1), (64, 0)])
def test_distances_for_row_lists(bits, metric):
    """szg_distances: the gather-by-row re-rank primitive, bit-equal to the oracle."""
    dim, n = 48, 3000
    rows = orc.synth_rows(17, 0, n, dim, bits)
    q = orc.synth_vectors(18, 0, 1, dim)[0]
    want = orc.all_distances(rows, dim, bits, metric, q)
    pick = np.array([0, 2999, 17, 17, 1500, 64, 63, 65], dtype=np.uint64)
    for devs in (None, [0, 0]):
        with ScanIndex(dim, bits, metric, devices=devs) as ix:
            ix.load(rows)
            got = ix.distances(q, pick)
            assert (got == want[pick.astype(int)]).all()
            with pytest.raises(Exception):
                ix.distances(q, [n])


@pytest.mark.parametrize("qpl", [1, 5, 16])
@pytest.mark.parametrize("bits,metric", [(4, 0), (4, 1), (8, 0), (8, 1), (16, 0), (32, 1), (64, 0)])
def test_query_major_launches(bits, metric, qpl):
    """One scan launch walks several queries' sweeps back to back ("queries_per_launch"):
    per-query constants of the integer paths, per-query filter masks and result slots
    must follow the query, whatever the grouping."""
    dim, n, k, nq = 40, 6000, 7, 37
    rows = orc.synth_rows(SEED + 300 + bits, 0, n, dim, bits)
    queries = orc.synth_vectors(SEED + 301, 0, nq, dim)
    queries[3] *= 40.0          # very different scales -> different qscale per query
    queries[4] *= 1e-3
    rng = np.random.default_rng(8)
    allow = rng.random((nq, n)) < 0.5
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        ix.set_option("multi_query", 0)
        ix.set_option("queries_per_launch", qpl)
        for masks in (None, allow):
            r, d, c = ix.search_topk(queries, k, allow=masks)
            for qi in range(nq):
                o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, queries[qi], k=k,
                                                     allow=None if masks is None else masks[qi].astype(np.uint8))
                assert_same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist)


@pytest.mark.parametrize("dim", [384, 768])
@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
def test_shape_specialised_kernels_match_any_shape(dim, metric):
    """4-bit rows of 384 / 768 dims take row-shape-specialised kernels (top-k and collect): the oracle's answer.  (The
    any-shape kernels serve every other dimension: test_topk_matches_oracle.)"""
    bits, n, k = 4, 30000, 10
    rows = orc.synth_rows(SEED + 400 + dim, 0, n, dim, bits)
    queries = orc.synth_vectors(SEED + 401, 0, 3, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        for _ in (1,):
            r, d, c = ix.search_topk(queries, k)
            for qi in range(3):
                o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, queries[qi], k=k)
                assert_same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist)
            radius = float(d[0, k - 1])
            rr, dd = ix.search_radius(queries[0], radius)
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, queries[0], radius=radius)
            assert_same(rr, dd, o_rows, o_dist)


def _check_all(ix, rows, dim, bits, metric, queries, k, allow=None):
    r, d, c = ix.search_topk(queries, k, allow=allow)
    for qi in range(queries.shape[0]):
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, queries[qi], k=k,
                                             allow=None if allow is None else allow[qi].astype(np.uint8))
        assert c[qi] == len(o_rows), (qi, c[qi], len(o_rows))
        assert_same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("mq", [0, 1])
def test_nan_among_the_first_k_rows_poisons_the_heap_like_the_reference(bits, mq):
    """consider() pushes the first k eligible rows whatever their distance (collection.go:608).
    A row antipodal (or parallel) to the query can come out of the unclamped acos (:831) as NaN;
    its scan key is the WORST possible, so it is nowhere near the candidates -- but sitting in
    the reference's heap it decides everything that follows.  Same for NaN / Inf elements.  The
    library must notice (exact distances of the first k eligible rows) and take the exact replay."""
    dim, n, k = 24, 4000, 6
    rng = np.random.default_rng(bits + mq)
    vecs = rng.uniform(-1, 1, (n, dim))
    queries = rng.uniform(-1, 1, (5, dim))
    # find a query scale whose antipode really yields NaN under the reference's arithmetic
    nan_q = None
    for t in range(400):
        q = rng.uniform(-1, 1, dim)
        row = orc.encode_rows(-q[None, :] * 0.5, bits)
        if np.isnan(orc.all_distances(row, dim, bits, SZG_COSINE, q)[0]):
            nan_q = q
            break
    assert nan_q is not None, "no antipodal NaN found (acos argument never left [-1, 1])"
    queries[0] = nan_q
    for where in (0, k - 1, k, 1500):       # inside the first k rows, at the edge, outside
        v = vecs.copy()
        v[where] = -nan_q * 0.5
        rows = orc.encode_rows(v, bits)
        with ScanIndex(dim, bits, SZG_COSINE) as ix:
            ix.load(rows)
            ix.set_option("multi_query", mq)
            _check_all(ix, rows, dim, bits, SZG_COSINE, queries, k)
            st = ix.stats()
            if where < k:
                assert st["full_replays"] >= 1
            # with a filter the "first k eligible" rows move: row `where` is the first allowed one
            allow = np.ones((queries.shape[0], n), dtype=bool)
            allow[:, :where] = False
            _check_all(ix, rows, dim, bits, SZG_COSINE, queries, k, allow=allow)
            # ... and a tombstone takes it out of the heap's history again
            ix.tombstone(where)
            alive = np.ones((queries.shape[0], n), dtype=bool)
            alive[:, where] = False
            r, d, c = ix.search_topk(queries, k)
            for qi in range(queries.shape[0]):
                o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, SZG_COSINE, queries[qi], k=k,
                                                     allow=alive[qi].astype(np.uint8))
                assert_same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist)


@pytest.mark.parametrize("metric", [SZG_EUCLIDEAN, SZG_COSINE])
@pytest.mark.parametrize("bits", [32, 64])
def test_nan_and_inf_elements_in_stored_rows(bits, metric):
    dim, n, k = 16, 3000, 5
    rng = np.random.default_rng(3)
    vecs = rng.uniform(-1, 1, (n, dim))
    queries = rng.uniform(-1, 1, (4, dim))
    for where, val in ((2, np.nan), (0, np.inf), (k - 1, -np.inf), (700, np.nan)):
        v = vecs.copy()
        v[where, 3] = val
        rows = orc.encode_rows(v, bits)
        for devs in (None, [0, 0]):
            with ScanIndex(dim, bits, metric, devices=devs) as ix:
                ix.load(rows)
                for mq in (0, 1):
                    ix.set_option("multi_query", mq)
                    _check_all(ix, rows, dim, bits, metric, queries, k)


@pytest.mark.parametrize("bits,metric", [(8, 1), (32, 0)])
def test_k_beyond_the_fused_selection(bits, metric):
    """The reference bounds K by nothing (collection.go:606-619): k past the LDS-resident lists
    is answered by the exact replay, not refused."""
    dim, n = 12, 9000
    rows = orc.synth_rows(SEED + 900 + bits, 0, n, dim, bits)
    queries = orc.synth_vectors(SEED + 901, 0, 2, dim)
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        for k in (5000, 8999, 12000):
            _check_all(ix, rows, dim, bits, metric, queries, k)
        assert ix.stats()["full_replays"] >= 6
        allow = (np.arange(n) % 3 != 0)[None, :].repeat(2, axis=0)
        _check_all(ix, rows, dim, bits, metric, queries, 5000, allow=allow)


@pytest.mark.parametrize("bits,metric,dim", [(4, 1, 3), (8, 0, 2), (32, 1, 16)])
def test_tie_mode_1_returns_a_valid_topk(bits, metric, dim):
    """tie_mode=1 trades the reference's heap-history order among EQUAL distances for speed (no
    exact replay): the answer must still be a correct top-k -- ascending, every distance the
    reference's float64 value for its row, and the same multiset of distances as the oracle's."""
    n, k = 1500, 12
    rng = np.random.default_rng(bits)
    vec = np.round(rng.uniform(-1, 1, (n, dim)) * 2) / 2        # coarse grid: many exactly equal distances
    rows = orc.encode_rows(vec, bits)
    Q = rng.uniform(-1, 1, (6, dim))
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        ix.set_option("tie_mode", 1)
        for mq in (0, 1):
            ix.set_option("multi_query", mq)
            r, d, c = ix.search_topk(Q, k)
            assert ix.stats()["full_replays"] == 0
            for qi in range(Q.shape[0]):
                want_all = orc.all_distances(rows, dim, bits, metric, Q[qi])
                _, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=k)
                got_r, got_d = r[qi, : c[qi]], d[qi, : c[qi]]
                assert c[qi] == len(o_dist) and len(set(int(x) for x in got_r)) == c[qi]
                assert (np.diff(got_d) >= 0).all()
                assert (got_d == want_all[got_r.astype(np.int64)]).all()
                assert (np.sort(got_d) == np.sort(np.asarray(o_dist))).all()


@pytest.mark.parametrize("bits", [32, 64])
def test_infinite_radius_with_zero_query(bits):
    """Radius = +inf (a distance of a row with an infinite element, used as the radius) and an all-zero
    query: the key threshold must not turn into NaN (0 * inf) -- every row at a non-NaN distance is a hit."""
    rng = np.random.default_rng(12)
    dim, n = 5, 40
    V = rng.uniform(-1, 1, (n, dim))
    V[3, 2] = np.inf
    V[7, 0] = np.nan
    rows = orc.encode_rows(V, bits)
    q = np.zeros(dim)
    with ScanIndex(dim, bits, SZG_EUCLIDEAN) as ix:
        ix.load(rows)
        r, d = ix.search_radius(q, np.inf)
    o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, SZG_EUCLIDEAN, q, radius=np.inf)
    assert len(o_rows) == n - 1
    assert [int(x) for x in r] == [int(x) for x in o_rows]
    assert (np.asarray(d) == np.asarray(o_dist)).all()


@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("n_inf", [3, 40])
def test_float32_rows_beyond_the_float32_norm_range(metric, n_inf):
    """Elements of 1e19 .. 3e37 are ordinary float32 values whose squared norm overflows float32 but not
    the reference's float64: such rows are forced into the candidate lists and ranked by the float64
    re-rank (they are the nearest rows here, by angle).  Rows with an Inf / NaN element take the same way
    in and are dropped on the host (NaN distance) unless they are among a query's first k rows -- also
    when there are more of them than list slots."""
    rng = np.random.default_rng(5)
    dim, n = 64, 5000
    V = rng.standard_normal((n, dim))
    Q = rng.standard_normal((20, dim))
    for i, s in enumerate([1e19, 1e20, 1e25, 1e30, 3e37]):
        V[100 + i] = Q[0] * s + rng.standard_normal(dim) * s * 1e-3
    V[200] = Q[1] * 1e-25                       # float32 norm underflows to 0
    for r in rng.choice(np.arange(300, n), n_inf, replace=False):
        V[r, int(rng.integers(0, dim))] = [np.inf, -np.inf, np.nan][int(r) % 3]
    rows = orc.encode_rows(V, 32)
    with ScanIndex(dim, 32, metric) as ix:
        ix.load(rows)
        for multi in (0, 1):
            ix.set_option("multi_query", multi)
            qs = Q if multi else Q[:3]
            r, d, c = ix.search_topk(qs, 10)
            for qi in range(qs.shape[0]):
                o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, metric, qs[qi], k=10)
                assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], (multi, qi)
                assert (d[qi, : c[qi]] == o_dist).all()


@pytest.mark.parametrize("multi", [0, 1])
def test_forced_rows_do_not_hide_the_true_nearest(multi):
    """k = 1 without slack: the one list slot goes to a forced row (float32 norm overflow, key -2), the true
    nearest row has key -1.  The escalation must start from the forced row's real key (from its float64
    distance), not from -2 (found by the fuzzer)."""
    rng = np.random.default_rng(9)
    dim, n = 5, 200
    V = rng.uniform(-1, 1, (n, dim))
    for r in (100, 120, 150, 180):
        V[r] *= 1e25
    rows = orc.encode_rows(V, 32)
    Q = rng.uniform(-1, 1, (16, dim))
    Q[0] = V[25]
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("slack", 0)
        ix.set_option("multi_query", multi)
        r, d, c = ix.search_topk(Q, 1)
        for qi in range(16):
            o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, SZG_COSINE, Q[qi], k=1)
            assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], qi
            assert (d[qi, : c[qi]] == o_dist).all()


def test_zero_query_with_a_forced_row_and_no_slack():
    """Fuzz seed 77, case 1111 (round 3): a zero cosine query is at distance exactly 1.0 from EVERY row
    (collection.go:828-830), so the reference keeps the first k visited.  With slack = 0 and k = 1 the candidate list
    holds one row -- a row with a NaN element is forced into it -- and nothing in it shows the tie; a zero query
    therefore always takes the exact replay."""
    dim, n = 64, 15
    rng = np.random.default_rng(3)
    vec = rng.uniform(-1, 1, (n, dim))
    vec[7, 5] = np.nan
    vec[9, 1] = np.inf
    rows = orc.encode_rows(vec, 32)
    for slack in (0, 16):
        with ScanIndex(dim, 32, SZG_COSINE) as ix:
            ix.load(rows)
            ix.set_option("slack", slack)
            ix.set_option("multi_query", 0)
            for k in (1, 3, 20):
                r, d, c = ix.search_topk(np.zeros(dim), k)
                o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, SZG_COSINE, np.zeros(dim), k=k)
                assert_same(r[0, : c[0]], d[0, : c[0]], o_rows, o_dist)
            if not os.environ.get("SZG_OPTIONS"):   # (the sketch pre-pass counts its own hand-over)
                assert ix.stats()["full_replays"] == 3
