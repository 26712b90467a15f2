"""The reference's own search tests (collection_test.go), re-stated against the
host mirror of its API: same calls, same assertions; the scan runs in HIP."""
import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import (Collection, CollectionOptions, Cosine, Euclidean, ScanIndex, SearchArgs)

pytestmark = pytest.mark.gpu


def test_exhaustive_search():
    """collection_test.go:549-612 TestExhaustiveSearch."""
    c = Collection(CollectionOptions(Name="test_exhaustive_search", DistanceMethod=Euclidean,
                                     DimensionCount=3))
    docs = [(1, [1.0, 2.0, 3.0], b"doc1"), (2, [4.0, 5.0, 6.0], b"doc2"), (3, [7.0, 8.0, 9.0], b"doc3")]
    for id, v, m in docs:
        c.AddDocument(id, v, m)
    res = c.Search(SearchArgs(Vector=[1.0, 2.0, 3.0], Precision="exact", K=3))
    assert len(res.Results) == len(docs)
    assert {r.ID for r in res.Results} == {1, 2, 3}
    assert res.PercentSearched == 100.0
    assert [r.Distance for r in res.Results] == [0.0, 5.196152422706632, 10.392304845413264]
    assert res.Results[0].Metadata == b"doc1"
    c.Close()


def test_collection_search_modes():
    """collection_test.go:283-382 TestCollectionSearch."""
    opts = CollectionOptions(Name="test_collection", DistanceMethod=Euclidean, DimensionCount=2)
    empty = Collection(opts)
    assert len(empty.Search(SearchArgs(Vector=[50, 50], K=5)).Results) == 0
    assert empty.Search(SearchArgs(Vector=[50, 50], K=5)).PercentSearched == 0
    empty.Close()

    c = Collection(CollectionOptions(Name="test_collection", DistanceMethod=Euclidean, DimensionCount=2))
    rng = np.random.default_rng(1)
    for i in range(10):
        c.AddDocument(i, rng.uniform(0, 100, 2), b"metadata")
    assert len(c.Search(SearchArgs(Vector=[50, 50], K=5)).Results) > 0           # Basic Search
    assert len(c.Search(SearchArgs(Vector=[50, 50], K=3)).Results) <= 3          # Max Count
    for r in c.Search(SearchArgs(Vector=[50, 50], Radius=10)).Results:           # Radius Search
        assert r.Distance <= 10
    res = c.Search(SearchArgs(Vector=[50, 50], K=5, Filter=lambda id, md: id % 2 == 0))
    assert len(res.Results) == 5 and all(r.ID % 2 == 0 for r in res.Results)     # Filter Function
    assert res.PercentSearched == 100.0   # counted before the filter (collection.go:589)
    c.Close()


def test_vector_search_with_4bit_quantization():
    """collection_test.go:614-667."""
    c = Collection(CollectionOptions(Name="4bit", DistanceMethod=Euclidean, DimensionCount=3,
                                     Quantization=4))
    rng = np.random.default_rng(2)
    vecs = rng.uniform(0, 1, (10, 3))
    for i in range(10):
        c.AddDocument(i, vecs[i], b"metadata")
    q = rng.uniform(0, 1, 3)
    res = c.Search(SearchArgs(Vector=q, K=5))
    assert len(res.Results) == 5
    rows = orc.encode_rows(vecs, 4)
    o_rows, o_dist, _ = orc.search_exact(rows, 3, 4, 0, q, k=5)
    assert [r.ID for r in res.Results] == [int(x) for x in o_rows]
    assert [r.Distance for r in res.Results] == list(o_dist)
    c.Close()


def test_cosine_exact_20000x3():
    """collection_test.go:23-103 (the exact half): 20 000 x 3, Cosine, K=10, query = doc 0."""
    rng = np.random.default_rng(0)
    vecs = rng.uniform(0, 1, (20000, 3))
    c = Collection(CollectionOptions(Name="cos", DistanceMethod=Cosine, DimensionCount=3))
    c.AddDocuments(range(20000), vecs, [b"metadata_%d" % i for i in range(20000)])
    res = c.Search(SearchArgs(Vector=vecs[0], K=10, Precision="exact"))
    o_rows, o_dist, _ = orc.search_exact(orc.encode_rows(vecs, 64), 3, 64, 1, vecs[0], k=10)
    assert [r.ID for r in res.Results] == [int(x) for x in o_rows]
    got = np.array([r.Distance for r in res.Results])
    assert ((got == o_dist) | (np.isnan(got) & np.isnan(o_dist))).all()
    assert res.PercentSearched == 100.0
    c.Close()


def test_add_update_remove_and_listing():
    """collection_test.go:459-534 / :196-281 in spirit: CRUD keeps the mirror in step."""
    c = Collection(CollectionOptions(Name="crud", DistanceMethod=Euclidean, DimensionCount=4,
                                     Quantization=32))
    for i in range(20):
        c.AddDocument(i, [i, 0, 0, 0], b"m%d" % i)
    assert c.GetDocumentCount() == 20
    assert list(c.GetDocument(7).Vector) == [7.0, 0, 0, 0]
    res = c.Search(SearchArgs(Vector=[7.2, 0, 0, 0], K=2, Precision="exact"))
    assert [r.ID for r in res.Results] == [7, 8]
    c.removeDocument(7)
    with pytest.raises(KeyError):
        c.GetDocument(7)
    res = c.Search(SearchArgs(Vector=[7.2, 0, 0, 0], K=2, Precision="exact"))
    assert [r.ID for r in res.Results] == [8, 6]
    assert res.PercentSearched == 100.0
    c.AddDocument(3, [7.1, 0, 0, 0], b"moved")          # rewriting an id replaces its vector
    res = c.Search(SearchArgs(Vector=[7.2, 0, 0, 0], K=1))
    assert res.Results[0].ID == 3 and res.Results[0].Metadata == b"moved"
    c.UpdateDocument(3, b"again")
    assert c.Search(SearchArgs(Vector=[7.2, 0, 0, 0], K=1)).Results[0].Metadata == b"again"
    # listing mode (K == 0 and Radius == 0): sorted *string* id order, Offset/Limit
    ids = [r.ID for r in c.Search(SearchArgs(Offset=0, Limit=5)).Results]
    assert ids == [0, 1, 10, 11, 12]
    ids = [r.ID for r in c.Search(SearchArgs(Offset=2, Limit=3)).Results]
    assert ids == [10, 11, 12]
    with pytest.raises(ValueError):
        c.AddDocument(99, [1, 2, 3], b"")                 # dimension mismatch panics (collection.go:432)
    with pytest.raises(ValueError):
        c.Search(SearchArgs(Vector=[1, 2, 3], K=1))       # query length is validated here
    c.Close()


def test_unsupported_options():
    with pytest.raises(ValueError):
        Collection(CollectionOptions(DistanceMethod=2, DimensionCount=3))
    with pytest.raises(ValueError):
        Collection(CollectionOptions(DistanceMethod=0, DimensionCount=3, Quantization=7))


@pytest.mark.parametrize("bits,metric", [(32, 1), (8, 0), (4, 1)])
def test_two_shards_on_one_device_equal_one_shard(bits, metric):
    """devices=[0, 0] exercises the in-process sharding + cross-shard assembly on one GPU."""
    dim, n = 40, 5000
    rows = orc.synth_rows(9, 0, n, dim, bits)
    Q = orc.synth_vectors(10, 0, 5, dim)
    allow = np.arange(n) % 5 != 0
    with ScanIndex(dim, bits, metric, devices=[0, 0]) as two, ScanIndex(dim, bits, metric) as one:
        two.load(rows)
        one.load(rows)
        assert two.rows == n
        for kw in ({}, {"allow": np.tile(allow, (5, 1))}):
            r2, d2, c2 = two.search_topk(Q, 10, **kw)
            r1, d1, c1 = one.search_topk(Q, 10, **kw)
            assert (r2 == r1).all() and (d2 == d1).all() and (c2 == c1).all()
        for qi in range(2):
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=10,
                                                 allow=allow.astype(np.uint8))
            assert [int(x) for x in r2[qi]] == [int(x) for x in o_rows]
            assert (d2[qi] == o_dist).all()
        alld = orc.all_distances(rows, dim, bits, metric, Q[0])
        radius = float(np.sort(alld)[40])
        ra, da = two.search_radius(Q[0], radius)
        rb, db = one.search_radius(Q[0], radius)
        assert (ra == rb).all() and (da == db).all() and len(ra) == 41
        assert (two.read_rows(0, n) == rows).all()
        two.tombstone(int(r2[0, 0]))
        r3, _, _ = two.search_topk(Q[0], 10)
        assert int(r2[0, 0]) not in [int(x) for x in r3[0]]


def test_concurrent_searches_from_threads():
    """The reference serves Searches concurrently under RLock (collection.go:570)."""
    import threading
    dim, n, bits = 32, 20000, 32
    rows = orc.synth_rows(21, 0, n, dim, bits)
    Q = orc.synth_vectors(22, 0, 24, dim)
    want = [orc.search_exact(rows, dim, bits, 1, Q[i], k=5)[0] for i in range(Q.shape[0])]
    with ScanIndex(dim, bits, 1) as ix:
        ix.load(rows)
        errs = []

        def worker(t):
            try:
                for i in range(t, Q.shape[0], 6):
                    r, _, _ = ix.search_topk(Q[i], 5)
                    assert [int(x) for x in r[0]] == [int(x) for x in want[i]]
            except Exception as e:  # pragma: no cover
                errs.append(e)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs


@pytest.mark.parametrize("bits,metric", [(64, 0), (32, 1), (8, 1), (4, 0)])
def test_collection_from_spanfile_end_to_end(tmp_path, bits, metric):
    """A SyzgyDB collection file -> C++ pager -> HBM mirror -> Search, against the
    oracle scanning the same records in the same (sorted-string) visit order."""
    import spanfile_writer as sw
    from syzgydb_amd import codec
    dim, n = 24, 700
    vecs = orc.synth_vectors(41, 0, n, dim)
    rows = codec.encode_rows(vecs, bits)
    ids = [3 * i + 1 for i in range(n)]
    docs = [(ids[i], b"doc-%d" % ids[i], rows[i].tobytes()) for i in range(n)]
    extra = [sw.span(9000, str(ids[5]), [(0, b"rewritten"), (1, rows[6].tobytes())])]
    path = tmp_path / "coll.dat"
    sw.collection_file(path, metric, dim, bits, docs, extra_spans=extra)
    rows = rows.copy()
    rows[5] = rows[6]
    order = orc.sorted_id_order(ids)
    visit_rows = rows[[int(i) for i in order]]
    c = Collection.from_spanfile(path)
    assert (c.DimensionCount, c.Quantization, c.DistanceMethod) == (dim, bits, metric)
    assert c.GetDocumentCount() == n
    for q in orc.synth_vectors(42, 0, 3, dim):
        res = c.Search(SearchArgs(Vector=q, K=7, Precision="exact"))
        o_rows, o_dist, _ = orc.search_exact(visit_rows, dim, bits, metric, q, k=7)
        assert [r.ID for r in res.Results] == [ids[int(order[int(x)])] for x in o_rows]
        assert [r.Distance for r in res.Results] == list(o_dist)
        assert res.PercentSearched == 100.0
    # a stored vector as the query: two identical records (5 was rewritten with 6's vector).
    # Under cosine the reference may produce NaN here (acos is not clamped, collection.go:831);
    # whatever it does, the answer has to be the oracle's.
    qs = orc.decode_vector(rows[5], dim, bits)
    hit = c.Search(SearchArgs(Vector=qs, K=2)).Results
    o_rows, o_dist, _ = orc.search_exact(visit_rows, dim, bits, metric, qs, k=2)
    assert [r.ID for r in hit] == [ids[int(order[int(x)])] for x in o_rows]
    got = np.array([r.Distance for r in hit])
    assert ((got == o_dist) | (np.isnan(got) & np.isnan(o_dist))).all()
    assert any(r.Metadata == b"rewritten" for r in c.Search(SearchArgs(Offset=0, Limit=n)).Results)
    c.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("bits,metric", [(4, 1), (8, 0), (16, 1), (32, 1), (64, 0)])
def test_pair_distances_and_average_distance(bits, metric):
    """szg_pair_distances == c.distance(doc1.Vector, doc2.Vector) bit for bit; the
    computeAverageDistance mirror (collection.go:348-400) sums them in pair order."""
    dim, n = 37, 600
    rows = orc.synth_rows(77 + bits, 0, n, dim, bits)
    c = Collection(CollectionOptions(Name="avg", DistanceMethod=metric, DimensionCount=dim,
                                     Quantization=bits), devices=[0, 0])
    c._index.load(rows)
    for i in range(n):
        c._row_of[1000 + i] = i
        c._id_of.append(1000 + i)
        c._meta.append(b"")
    rng = np.random.default_rng(5)
    a = rng.integers(0, n, 300).astype(np.uint64)
    b = rng.integers(0, n, 300).astype(np.uint64)
    a[:5] = b[:5]                        # self pairs
    got = c._index.pair_distances(a, b)
    vec = [orc.decode_vector(rows[i], dim, bits) for i in range(n)]
    fn = orc.angular if metric == 1 else orc.euclidean
    want = np.array([fn(vec[int(x)], vec[int(y)]) for x, y in zip(a, b)])
    assert np.array_equal(got, want, equal_nan=True)

    draws = [int(x) for x in rng.integers(0, n, 200)]
    it = iter(draws)
    avg = c.computeAverageDistance(100, intn=lambda m: next(it))
    total, count = 0.0, 0
    for i in range(100):
        r1, r2 = draws[2 * i], draws[2 * i + 1]
        if r1 == r2:
            continue
        total += fn(vec[r1], vec[r2])
        count += 1
    assert avg == total / count
    assert c.computeAverageDistance(0) == 0.0
    c.Close()


def test_append_on_two_shards_keeps_filter_bits_aligned():
    """Rows appended to a two-shard handle after a small load: every shard must still start at a
    multiple of 64 rows, or the per-shard slices of the filter / tombstone bitmaps shift
    (found by scripts/fuzz_gpu.py)."""
    dim, bits, metric, n = 24, 32, 0, 300
    rows = orc.synth_rows(901, 0, n, dim, bits)
    Q = orc.synth_vectors(902, 0, 5, dim)
    rng = np.random.default_rng(4)
    allow = rng.random((5, n)) < 0.5
    for split in (1, 5, 64, 70, 299):
        with ScanIndex(dim, bits, metric, devices=[0, 0]) as ix:
            ix.load(rows[:split])
            ix.append(rows[split:])
            ix.tombstone(3)
            assert (ix.read_rows(0, n) == rows).all()
            r, d, c = ix.search_topk(Q, 10, allow=allow)
            for qi in range(5):
                m = allow[qi].copy()
                m[3] = False
                o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=10, allow=m.astype(np.uint8))
                assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], (split, qi)
                assert (d[qi, : c[qi]] == o_dist).all()


def test_filter_bitmask_cache():
    """A filter's verdicts are kept per (filter, collection version): the second search with the
    same filter does not call it again; any mutation invalidates."""
    dim = 8
    c = Collection(CollectionOptions(Name="f", DistanceMethod=Euclidean, DimensionCount=dim, Quantization=32))
    rng = np.random.default_rng(1)
    for i in range(50):
        c.AddDocument(i, rng.uniform(-1, 1, dim), b"m%d" % i)
    calls = [0]

    def even(id, meta):
        calls[0] += 1
        return id % 2 == 0

    q = rng.uniform(-1, 1, dim)
    a = c.Search(SearchArgs(Vector=q, K=5, Filter=even, Precision="exact"))
    assert calls[0] == 50 and all(r.ID % 2 == 0 for r in a.Results)
    b = c.Search(SearchArgs(Vector=q, K=5, Filter=even, Precision="exact"))
    assert calls[0] == 50 and [r.ID for r in b.Results] == [r.ID for r in a.Results]
    c.AddDocument(100, q, b"exact match")          # version changes: verdicts are recomputed
    d = c.Search(SearchArgs(Vector=q, K=5, Filter=even, Precision="exact"))
    assert calls[0] == 101 and d.Results[0].ID == 100
    # a named filter survives a new closure object
    e = c.Search(SearchArgs(Vector=q, K=5, Filter=lambda i, m: i % 2 == 0, FilterKey="even", Precision="exact"))
    f = c.Search(SearchArgs(Vector=q, K=5, Filter=lambda i, m: 1 / 0, FilterKey="even", Precision="exact"))
    assert [r.ID for r in f.Results] == [r.ID for r in e.Results] == [r.ID for r in d.Results]
    c.Close()


@pytest.mark.parametrize("bits,metric,devices", [(32, 1, [0]), (8, 0, [0]), (4, 1, [0, 0]), (32, 0, [0, 0])])
def test_concurrent_mixed_batches_from_threads(bits, metric, devices):
    """Eight threads on one handle, each mixing single queries, small and large batches (one sweep
    per query, shared sweeps, radius searches, filter masks): every answer is the oracle's."""
    import threading
    dim, n = 64, 30000
    rows = orc.synth_rows(31 + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(32, 0, 120, dim)
    want = [orc.search_exact(rows, dim, bits, metric, Q[i], k=7) for i in range(Q.shape[0])]
    allow = (np.arange(n) % 3 != 0)
    want_f = [orc.search_exact(rows, dim, bits, metric, Q[i], k=7, allow=allow.astype(np.uint8)) for i in range(8)]
    with ScanIndex(dim, bits, metric, devices=devices) as ix:
        ix.load(rows)
        errs = []

        def worker(t):
            try:
                rng = np.random.default_rng(t)
                for it in range(12):
                    nq = int(rng.choice([1, 1, 2, 5, 16, 40]))
                    lo = int(rng.integers(0, Q.shape[0] - nq))
                    r, d, c = ix.search_topk(Q[lo:lo + nq], 7)
                    for j in range(nq):
                        assert [int(x) for x in r[j, : c[j]]] == [int(x) for x in want[lo + j][0]], (t, it, j)
                        assert (d[j, : c[j]] == want[lo + j][1]).all()
                    if it % 4 == 0:
                        i = int(rng.integers(0, 8))
                        r, d, c = ix.search_topk(Q[i], 7, allow=allow)
                        assert [int(x) for x in r[0, : c[0]]] == [int(x) for x in want_f[i][0]]
                        rr, dd = ix.search_radius(Q[i], float(want[i][1][3]))
                        assert [int(x) for x in rr] == [int(x) for x in want[i][0][:len(rr)]] and len(rr) >= 4
            except Exception as e:  # pragma: no cover
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[:3]


def test_concurrent_single_queries_are_coalesced():
    """Many threads, one query each (the reference's concurrent Searches under RLock): callers that
    arrive while a batch is in flight are answered together by one shared sweep; answers unchanged."""
    import threading
    dim, n, bits, k = 48, 200000, 32, 6
    rows = orc.synth_rows(41, 0, n, dim, bits)
    Q = orc.synth_vectors(42, 0, 96, dim)
    want = [orc.search_exact(rows, dim, bits, 1, Q[i], k=k) for i in range(Q.shape[0])]
    rng = np.random.default_rng(3)
    allow = rng.random((4, n)) < 0.3
    want_f = {i: orc.search_exact(rows, dim, bits, 1, Q[i], k=k, allow=allow[i % 4].astype(np.uint8))
              for i in range(0, Q.shape[0], 3)}
    with ScanIndex(dim, bits, 1) as ix:
        ix.load(rows)
        errs = []

        def worker(t):
            try:
                for i in range(t, Q.shape[0], 16):
                    kk = k if i % 5 else k - 1        # a second k in the mix: batches are per k
                    if i % 3 == 0:                    # every third caller brings its own filter
                        r, d, c = ix.search_topk(Q[i], kk, allow=allow[i % 4])
                        w = want_f[i]
                    else:
                        r, d, c = ix.search_topk(Q[i], kk)
                        w = want[i]
                    assert [int(x) for x in r[0, : c[0]]] == [int(x) for x in w[0][:kk]], i
                    assert (d[0, : c[0]] == w[1][:kk]).all()
            except Exception as e:  # pragma: no cover
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[:3]
        assert ix.stats()["queries"] == 96
        ix.set_option("coalesce", 0)
        ix.reset_stats()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs and ix.stats()["mq_queries"] == 0

    # that sweeps really are shared: a corpus whose sweep takes long enough (~100 us) for 32
    # threads to pile up behind the leader; answers must equal the one-call-at-a-time ones
    from syzgydb_amd.synth import synth_vectors
    dim, n = 128, 1_500_000
    Q = synth_vectors(43, 0, 64, dim)
    with ScanIndex(dim, 32, 1) as ix:
        ix.synth(n, 44)
        ix.set_option("coalesce", 0)
        base = [ix.search_topk(Q[i], k)[0][0].copy() for i in range(Q.shape[0])]
        ix.set_option("coalesce", 1)
        ix.reset_stats()
        start = threading.Barrier(32)
        bad = []

        def hammer(t):
            start.wait()
            for rep in range(10):
                i = (t * 7 + rep * 3) % Q.shape[0]
                r, _, _ = ix.search_topk(Q[i], k)
                if not (r[0] == base[i]).all():
                    bad.append(i)
        th = [threading.Thread(target=hammer, args=(t,)) for t in range(32)]
        [t.start() for t in th]
        [t.join() for t in th]
        st = ix.stats()
        assert not bad and st["queries"] == 320
        assert st["mq_queries"] > 0          # some callers shared a sweep


def test_tie_order_follows_sorted_string_ids_after_appends():
    """The reference's deterministic exact scan visits records in sort.Strings order of their
    decimal ids (spanfile.go:540-560) and the first visited wins a tie at the k boundary
    (collection.go:608).  Documents added later, in any id order, must not change that: the
    mirror appends an id that sorts last and re-pages itself in order otherwise (the rule
    go/syzgy_gpu.go follows too)."""
    dim, bits = 4, 8
    rng = np.random.default_rng(11)
    base = rng.uniform(-1, 1, (3, dim))
    ids = [5, 40, 100, 2, 31, 7, 1000, 12, 3, 64, 9, 77, 8, 200, 30, 6]     # "100" < "2" < "31" ...
    vec = {id: base[i % 3] for i, id in enumerate(ids)}                      # many exactly equal rows
    q = base[1] + 0.01
    for metric in (Euclidean, Cosine):
        c = Collection(CollectionOptions(Name="ties", DistanceMethod=metric, DimensionCount=dim,
                                         Quantization=bits))
        loaded = []
        for step, id in enumerate(ids):
            c.AddDocument(id, vec[id], b"m%d" % id)
            loaded.append(id)
            if step in (2, 7, len(ids) - 1):
                order = sorted(loaded, key=str)                                # the reference's visit order
                rows = orc.encode_rows(np.stack([vec[i] for i in order]), bits)
                for k in (1, 2, 4, 5):
                    want_rows, want_d, _ = orc.search_exact(rows, dim, bits, metric, q, k=k)
                    got = c.Search(SearchArgs(Vector=q, K=k, Precision="exact"))
                    assert [r.ID for r in got.Results] == [order[int(r)] for r in want_rows], (metric, step, k)
                    assert [r.Distance for r in got.Results] == list(want_d)
        # removal keeps the order of the rest; a re-add of the removed id lands in its sorted place
        c.removeDocument(31)
        c.AddDocument(31, vec[31], b"again")
        order = sorted(loaded, key=str)
        rows = orc.encode_rows(np.stack([vec[i] for i in order]), bits)
        want_rows, want_d, _ = orc.search_exact(rows, dim, bits, metric, q, k=6)
        got = c.Search(SearchArgs(Vector=q, K=6, Precision="exact"))
        assert [r.ID for r in got.Results] == [order[int(r)] for r in want_rows]
        c.Close()


def test_collection_mirror_with_the_sketch_option():
    """Collection(..., sketch=True): lone Searches on a float32 collection go through the 8-bit pre-pass;
    ids, order and distances as without it."""
    rng = np.random.default_rng(21)
    dim, n = 32, 6000
    V = rng.standard_normal((n, dim))
    a = Collection(CollectionOptions(Name="a", DistanceMethod=Cosine, DimensionCount=dim, Quantization=32))
    b = Collection(CollectionOptions(Name="b", DistanceMethod=Cosine, DimensionCount=dim, Quantization=32), sketch=True)
    ids = list(range(1, n + 1))
    a.AddDocuments(ids, V, [b"m%d" % i for i in ids])
    b.AddDocuments(ids, V, [b"m%d" % i for i in ids])
    for qi in range(6):
        q = rng.standard_normal(dim)
        ra = a.Search(SearchArgs(Vector=q, K=7, Precision="exact"))
        rb = b.Search(SearchArgs(Vector=q, K=7, Precision="exact"))
        assert [(r.ID, r.Distance, r.Metadata) for r in ra.Results] == [(r.ID, r.Distance, r.Metadata) for r in rb.Results]
    st = b._index.stats()
    assert st["sketch_queries"] + st["sketch_fallbacks"] == 6
    b.removeDocument(ra.Results[0].ID)
    a.removeDocument(ra.Results[0].ID)
    ra = a.Search(SearchArgs(Vector=q, K=7, Precision="exact"))
    rb = b.Search(SearchArgs(Vector=q, K=7, Precision="exact"))
    assert [(r.ID, r.Distance) for r in ra.Results] == [(r.ID, r.Distance) for r in rb.Results]
    a.Close()
    b.Close()


def test_ascending_integer_ids_do_not_repage_the_mirror():
    """"10" < "9" as strings, so plain ascending ids 1, 2, 3, ... are out of sort.Strings order from
    the tenth on.  A new document is still appended in place; the mirror is re-paged in the reference's
    deterministic order only when an answer really depends on the visit order (a tie among the best k+1), and in
    production mode (strict_order=False: the reference iterates a Go map in random order) never."""
    dim, bits, metric = 16, 32, Cosine
    rng = np.random.default_rng(5)
    V = rng.standard_normal((300, dim))
    for strict in (True, False):
        c = Collection(CollectionOptions(Name="asc", DistanceMethod=metric, DimensionCount=dim, Quantization=bits),
                       strict_order=strict)
        for i in range(200):
            c.AddDocument(i + 1, V[i], b"")
            if i % 20 == 19:   # add / search interleaved
                q = rng.standard_normal(dim)
                got = c.Search(SearchArgs(Vector=q, K=5, Precision="exact"))
                order = sorted(range(1, i + 2), key=str)
                rows = orc.encode_rows(np.stack([V[j - 1] for j in order]), bits)
                want_rows, want_d, _ = orc.search_exact(rows, dim, bits, metric, q, k=5)
                assert [r.ID for r in got.Results] == [order[int(r)] for r in want_rows]   # no ties: any order agrees
                assert [r.Distance for r in got.Results] == list(want_d)
        assert c.resorts == 0, c.resorts
        # now a real tie at the k boundary: ids 150 and 30 hold the same vector, k = 1 -- the reference's
        # deterministic scan visits "150" before "30" (string order) and keeps the first
        c.AddDocument(30, V[149], b"")
        q = V[149] + 1e-3
        got = c.Search(SearchArgs(Vector=q, K=1, Precision="exact"))
        if strict:
            assert c.resorts == 1 and got.Results[0].ID == 150
            c.AddDocument(201, V[200], b"")          # "201" < "99": stale again, but no tie -> no re-page
            c.Search(SearchArgs(Vector=rng.standard_normal(dim), K=3, Precision="exact"))
            assert c.resorts == 1
        else:
            assert c.resorts == 0 and got.Results[0].ID in (30, 150)   # either is a reference answer
        c.Close()


def test_search_batch_equals_search_by_search():
    """Collection.SearchBatch (the Go binding's searchExactBatch): one library call for a list of Searches --
    top-k with one K through a shared sweep, radius Searches through szg_search_radius_batch -- each with its own
    Filter; results identical to Search called one by one, and to the oracle."""
    dim, bits, n = 48, 8, 5000
    rng = np.random.default_rng(9)
    V = rng.uniform(-1, 1, (n, dim))
    for metric in (Cosine, Euclidean):
        c = Collection(CollectionOptions(Name="batch", DistanceMethod=metric, DimensionCount=dim, Quantization=bits))
        ids = list(range(1000, 1000 + n))
        c.AddDocuments(ids, V, [b"m%d" % (i % 7) for i in ids])
        Q = rng.uniform(-1, 1, (21, dim))
        flt = [None, (lambda id, meta: meta == b"m3"), (lambda id, meta: id % 2 == 0)]
        args = [SearchArgs(Vector=Q[i], K=6, Precision="exact", Filter=flt[i % 3]) for i in range(len(Q))]
        before = c._index.stats()["mq_queries"]
        got = c.SearchBatch(args)
        assert c._index.stats()["mq_queries"] - before == len(Q)      # one shared-sweep call for all of them
        for a, g in zip(args, got):
            one = c.Search(a)
            assert [(r.ID, r.Distance, r.Metadata) for r in g.Results] == [(r.ID, r.Distance, r.Metadata) for r in one.Results]
            assert g.PercentSearched == one.PercentSearched == 100.0
        # radius: every query its own radius (its 9th neighbour's distance)
        rad = [c.Search(SearchArgs(Vector=Q[i], K=9, Precision="exact")).Results[-1].Distance for i in range(8)]
        rargs = [SearchArgs(Vector=Q[i], Radius=rad[i], Precision="exact", Filter=flt[i % 3]) for i in range(8)]
        for a, g in zip(rargs, c.SearchBatch(rargs)):
            one = c.Search(a)
            assert [(r.ID, r.Distance) for r in g.Results] == [(r.ID, r.Distance) for r in one.Results]
            assert all(r.Distance <= a.Radius for r in g.Results)
        order = sorted(ids, key=str)                               # the oracle on the reference's visit order
        rows = orc.encode_rows(np.stack([V[i - 1000] for i in order]), bits)
        want_rows, want_d, _ = orc.search_exact(rows, dim, bits, metric, Q[0], k=6)
        assert [r.ID for r in got[0].Results] == [order[int(r)] for r in want_rows]
        assert [r.Distance for r in got[0].Results] == list(want_d)
        # mixed kinds fall back to Search by Search
        mixed = c.SearchBatch([args[0], rargs[0], SearchArgs(Offset=0, Limit=3)])
        assert len(mixed) == 3 and len(mixed[2].Results) == 3
        c.Close()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("bits,devices", [(8, [0]), (32, [0, 0])])
def test_concurrent_long_calls_with_finisher_threads(bits, devices):
    """Four threads on one handle, each issuing calls of 200-400 queries (3+ shared-sweep batches: producer + finisher
    thread per call, all of them competing for the shards' four contexts) mixed with radius batches: no deadlock,
    every answer the oracle's."""
    import threading
    dim, n, metric = 48, 6000, 1
    rows = orc.synth_rows(71 + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(72, 0, 400, dim)
    want = {i: orc.search_exact(rows, dim, bits, metric, Q[i], k=6) for i in range(0, 400, 23)}
    with ScanIndex(dim, bits, metric, devices=devices) as ix:
        ix.load(rows)
        _, dd, _ = ix.search_topk(Q[:40], 30)
        radii = dd[:, -1].copy()
        want_r = {i: orc.search_exact(rows, dim, bits, metric, Q[i], radius=float(radii[i])) for i in (0, 17, 39)}
        errs = []

        def worker(t):
            try:
                for it in range(3):
                    nq = (200, 330, 400)[(t + it) % 3]
                    r, d, c = ix.search_topk(Q[:nq], 6)
                    for i in want:
                        if i < nq:
                            assert [int(x) for x in r[i, : c[i]]] == [int(x) for x in want[i][0]], (t, it, i)
                            assert (d[i, : c[i]] == want[i][1]).all()
                    hits = ix.search_radius_batch(Q[:40], radii)
                    for i in want_r:
                        assert [int(x) for x in hits[i][0]] == [int(x) for x in want_r[i][0]], (t, it, i)
            except Exception as e:  # pragma: no cover
                errs.append(e)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
