"""SURVEY.md 8f-3 on the GPU: the default ("medium") search path with its candidates scored in
bulk by szg_distances.  The host walk of syzgydb_amd/lsh.py (the reference's forest and queue,
windows of leaves scored in one device call each) must return exactly what lshTree.search +
consider() return on the same forest: rows, order, float64 distances, pointsSearched."""
import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import ScanIndex, lsh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,metric,dim", [(32, 1, 96), (8, 0, 48), (4, 1, 64), (64, 0, 16), (16, 1, 32)])
def test_lsh_bulk_rerank_equals_reference_walk(bits, metric, dim):
    n = 6000
    rows = orc.synth_rows(700 + bits, 0, n, dim, bits)
    forest_o = orc.LshForest(rows, dim, bits, metric, threshold=100, num_trees=5, seed=11)   # collection.go:292
    forest = lsh.LshForest(metric=metric, **forest_o.export())
    Q = orc.synth_vectors(701, 0, 4, dim)
    Q[1] = orc.decode_vector(rows[1234], dim, bits)          # a stored vector as the query
    rng = np.random.default_rng(5)
    allow = rng.random(n) < 0.5
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        for qi in range(Q.shape[0]):
            for k, radius, flt in ((10, 0.0, None), (100, 0.0, allow), (0, 0.46 if metric else 0.9 * np.sqrt(dim / 6), None)):
                er, ed, es, _ = forest_o.search(Q[qi], k=k, radius=radius, allow=flt)
                r, d, s = lsh.search(forest, ix, Q[qi], k=k, radius=radius, allow=flt)
                assert list(map(int, r)) == list(map(int, er)), (qi, k, radius)
                assert ((d == ed) | (np.isnan(d) & np.isnan(ed))).all()
                assert s == es   # pointsSearched (the Euclidean forests of the reference prune little: s may reach n)
        # few device calls per search: windows of ~2048 candidates, not one call per candidate or leaf
        calls = []
        real = ix.distances
        ix.distances = lambda q, ids: (calls.append(len(ids)), real(q, ids))[1]
        _, _, s = lsh.search(forest, ix, Q[0], k=10)
        assert len(calls) <= 2 + s // 1024 and sum(calls) >= s


def test_cosine_distance_precision_comparison_20000x3():
    """The ONLY thing the reference's tests hold about its medium path -- TestCosineDistancePrecisionComparison,
    collection_test.go:23-103 -- restated on the GPU path: 20 000 x 3 (Quantization defaults to 64,
    collection.go:254-256), Cosine, vectors in [0,1)^3, query = document 0, K = 10, Precision "exact" vs the
    default "medium" (the forest walk, its candidates scored in bulk by szg_distances).  The reference asserts:
    the same number of results, |d_exact - d_medium| / d_exact <= 1 position by position, PercentSearched < 100.
    (Go's seeded math/rand stream is not reproducible here: the vectors come from numpy, the forest from the
    oracle's builder, collection.go:292's parameters.)"""
    n, dim, bits, metric, k = 20000, 3, 64, 1, 10
    rng = np.random.default_rng(0)
    vecs = rng.uniform(0, 1, (n, dim))
    rows = orc.encode_rows(vecs, bits)
    forest_o = orc.LshForest(rows, dim, bits, metric, threshold=100, num_trees=5, seed=0)
    forest = lsh.LshForest(metric=metric, **forest_o.export())
    q = vecs[0]
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        er, ed, ec = ix.search_topk(q, k)                       # Precision "exact"
        mr, md, searched = lsh.search(forest, ix, q, k=k)       # Precision "medium"
    assert ec[0] == k and len(mr) == ec[0]                      # collection_test.go:82-84
    # document 0 itself: the unclamped acos makes its distance 0 or NaN (collection.go:831); the reference's
    # ratio test is vacuous there (0/0), as it is in the Go test
    for i in range(k):
        de, dm = float(ed[0, i]), float(md[i])
        if de > 0 and not (np.isnan(de) or np.isnan(dm)):
            assert abs(de - dm) / de <= 1.0, (i, de, dm)        # :87-95
    percent = float(searched) / float(n) * 100.0                # collection.go:700-709
    assert 0 < percent < 100.0, percent                         # collection_test.go:96-98
    # and the medium answer is exactly the reference walk's on this forest
    wr, wd, ws, _ = forest_o.search(q, k=k)
    assert list(map(int, mr)) == list(map(int, wr)) and ((md == wd) | (np.isnan(md) & np.isnan(wd))).all()
    assert searched == ws
