"""The 8-bit sketch pre-pass ("sketch" option, float32 cosine collections): a sweep of a quarter of the bytes
picks the candidates, the float32 rows and float64 arithmetic decide -- answers identical to the reference loop's."""
import os

import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import ScanIndex, SZG_COSINE, SZG_EUCLIDEAN

pytestmark = pytest.mark.gpu
DEFAULT_TUNABLES = not os.environ.get("SZG_OPTIONS")


def check(ix, rows, dim, Q, k, allow=None, live=None, metric=SZG_COSINE):
    kw = {}
    if allow is not None:
        kw["allow"] = np.tile(allow, (Q.shape[0], 1))
    r, d, c = ix.search_topk(Q, k, **kw)
    m = None
    if allow is not None or live is not None:
        m = np.ones(rows.shape[0], bool)
        if allow is not None:
            m &= allow
        if live is not None:
            m &= live
        m = m.astype(np.uint8)
    for qi in range(Q.shape[0]):
        o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, metric, Q[qi], k=k, allow=m)
        assert c[qi] == len(o_rows)
        assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], qi
        got, want = d[qi, : c[qi]], o_dist
        assert ((got == want) | (np.isnan(got) & np.isnan(want))).all(), qi


@pytest.mark.parametrize("dim,n,k", [(128, 30000, 10), (768, 8000, 10), (100, 20000, 3), (48, 40000, 30)])
@pytest.mark.parametrize("multi", [0, 1])
def test_sketch_prepass_matches_oracle(dim, n, k, multi):
    rows = orc.synth_rows(700 + dim, 0, n, dim, 32)
    Q = orc.synth_vectors(701 + dim, 0, 24, dim)
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("sketch", 1)
        ix.set_option("multi_query", multi)
        check(ix, rows, dim, Q, k)
        st = ix.stats()
        if multi:   # a batch shares one sweep on the matrix cores: cheaper per query than any pre-pass
            assert st["sketch_queries"] + st["sketch_fallbacks"] == 0 or not DEFAULT_TUNABLES
            r1, d1, c1 = ix.search_topk(Q[0], k)          # a lone query takes the pre-pass
            o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, SZG_COSINE, Q[0], k=k)
            assert [int(x) for x in r1[0, : c1[0]]] == [int(x) for x in o_rows] and (d1[0, : c1[0]] == o_dist).all()
            assert ix.stats()["sketch_queries"] + ix.stats()["sketch_fallbacks"] >= 1
        else:
            assert st["sketch_queries"] + st["sketch_fallbacks"] == 24
            if DEFAULT_TUNABLES:
                assert st["sketch_queries"] >= 20   # random rows: the pre-pass settles (nearly) every query
        ix.set_option("sketch", 0)
        ix.reset_stats()
        check(ix, rows, dim, Q[:4], k)
        assert ix.stats()["sketch_queries"] == 0


def test_sketch_follows_mutations_masks_and_odd_rows():
    """Appends, overwrites and tombstones after the first search; filter masks; zero rows, rows with
    Inf / NaN elements, duplicates (equal distances: the float32 path's exact replay answers), rows far
    beyond the float32 norm range."""
    rng = np.random.default_rng(31)
    dim, n = 64, 12000
    V = rng.standard_normal((n + 3000, dim))
    V[5] = 0.0
    V[6, 3] = np.inf
    V[7, 9] = np.nan
    V[50] = V[51]
    V[60] *= 1e25
    rows = orc.encode_rows(V, 32)
    Q = rng.standard_normal((12, dim))
    Q[3] = V[50] + rng.standard_normal(dim) * 1e-6     # its two best rows are exact duplicates
    allow = rng.random(n + 3000) < 0.4
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows[:n])
        ix.set_option("sketch", 1)
        ix.set_option("multi_query", 0)
        check(ix, rows[:n], dim, Q, 10)
        check(ix, rows[:n], dim, Q, 10, allow=allow[:n])
        ix.append(rows[n : n + 2000])                   # sketched incrementally at the next search
        ix.append_vectors(V[n + 2000 :])
        check(ix, rows, dim, Q, 10)
        for r0 in (10, 11000, n + 2500):                 # overwrite: the row's sketch is rebuilt
            V[r0] = Q[1] * 3.0 + rng.standard_normal(dim) * 0.01
            rows[r0] = orc.encode_rows(V[r0].reshape(1, -1), 32)[0]
            ix.overwrite(r0, rows[r0])
        check(ix, rows, dim, Q, 10)
        live = np.ones(n + 3000, bool)
        for r0 in (10, 11000, 77):
            ix.tombstone(r0)
            live[r0] = False
        check(ix, rows, dim, Q, 10, live=live)
        check(ix, rows, dim, Q, 10, allow=allow, live=live)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            assert st["sketch_queries"] > 0 and st["sketch_fallbacks"] > 0


def test_sketch_clustered_rows():
    """Tight clusters (rows within 1e-3 of each other in angle, far inside the sketch's own error): the bound
    D - A cannot separate them, those queries fall back -- the answers stay exact either way."""
    rng = np.random.default_rng(32)
    dim, n = 96, 16000
    centers = rng.standard_normal((40, dim))
    V = centers[rng.integers(0, 40, n)] + rng.standard_normal((n, dim)) * 1e-3
    rows = orc.encode_rows(V, 32)
    Q = centers[:10] + rng.standard_normal((10, dim)) * 1e-3
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("sketch", 1)
        ix.set_option("multi_query", 0)
        check(ix, rows, dim, Q, 10)
        st = ix.stats()
        assert st["sketch_queries"] + st["sketch_fallbacks"] == 10


def test_sketch_concurrent_lone_queries():
    """Eight threads, one query per call, no coalescing: pre-pass searches run side by side after the sync."""
    import threading
    dim, n, k = 64, 30000, 5
    rows = orc.synth_rows(740, 0, n, dim, 32)
    Q = orc.synth_vectors(741, 0, 64, dim)
    want = [orc.search_exact(rows, dim, 32, SZG_COSINE, Q[i], k=k) for i in range(Q.shape[0])]
    bad = []
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("sketch", 1)
        ix.set_option("coalesce", 0)

        def work(t):
            for i in range(t, Q.shape[0], 8):
                r, d, c = ix.search_topk(Q[i], k)
                if [int(x) for x in r[0, : c[0]]] != [int(x) for x in want[i][0]] or not (d[0, : c[0]] == want[i][1]).all():
                    bad.append(i)

        th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        st = ix.stats()
        assert not bad, bad
        assert st["sketch_queries"] + st["sketch_fallbacks"] == 64


@pytest.mark.parametrize("dim,n,k", [(128, 30000, 10), (768, 8000, 25), (100, 20000, 3)])
def test_sketch_prepass_euclidean(dim, n, k):
    """Euclidean collections: one scale for the whole collection (the largest |x_i|), the bound is the largest
    Euclidean distance row <-> sketch; a later row beyond the scale forces a rebuild."""
    rng = np.random.default_rng(50 + dim)
    V = rng.uniform(-1, 1, (n + 500, dim)) * 3.0 + 0.5
    rows = orc.encode_rows(V, 32)
    Q = rng.uniform(-1, 1, (16, dim)) * 3.0 + 0.5
    with ScanIndex(dim, 32, SZG_EUCLIDEAN) as ix:
        ix.load(rows[:n])
        ix.set_option("sketch", 1)
        ix.set_option("multi_query", 0)
        check(ix, rows[:n], dim, Q, k, metric=SZG_EUCLIDEAN)
        st = ix.stats()
        assert st["sketch_queries"] + st["sketch_fallbacks"] == 16
        if DEFAULT_TUNABLES:
            assert st["sketch_queries"] >= 12
        V[n + 7] *= 40.0                                 # beyond the collection's scale: rebuild with the new one
        V[n + 9, 2] = np.inf
        rows = orc.encode_rows(V, 32)
        ix.append(rows[n:])
        check(ix, rows, dim, Q, k, metric=SZG_EUCLIDEAN)
        ix.tombstone(n + 7)
        live = np.ones(n + 500, bool)
        live[n + 7] = False
        check(ix, rows, dim, Q, k, live=live, metric=SZG_EUCLIDEAN)
