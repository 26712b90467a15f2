"""The shared (multi-query, MFMA) sweep against the oracle: same bar as the
single-query path -- ids identical, float64 distances bit-equal."""
import os

import numpy as np
import pytest

import oracle as orc
from syzgydb_amd import ScanIndex, SZG_COSINE, SZG_EUCLIDEAN

pytestmark = pytest.mark.gpu

# statistics of a particular path are asserted only under the default tunables (scripts/test_option_sweep.sh
# re-runs this file with SZG_OPTIONS set: the ANSWERS must not change, the path taken may)
DEFAULT_TUNABLES = not os.environ.get("SZG_OPTIONS")
TIE_EXACT = "tie_mode=1" not in os.environ.get("SZG_OPTIONS", "")


def check(ix, rows, dim, Q, k, allow=None, bits=32, metric=1):
    kw = {}
    if allow is not None:
        kw["allow"] = np.tile(allow, (Q.shape[0], 1))
    r, d, c = ix.search_topk(Q, k, **kw)
    for qi in range(Q.shape[0]):
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=k,
                                             allow=None if allow is None else allow.astype(np.uint8))
        assert c[qi] == len(o_rows)
        assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], qi
        got, want = d[qi, : c[qi]], o_dist
        assert ((got == want) | (np.isnan(got) & np.isnan(want))).all(), qi


@pytest.mark.parametrize("dim,n", [(768, 3000), (384, 5000), (128, 20000), (20, 1500), (3, 700), (100, 40)])
@pytest.mark.parametrize("nq", [32, 20, 9, 70, 101])
def test_shared_sweep_matches_oracle(dim, n, nq):
    rows = orc.synth_rows(31 + dim, 0, n, dim, 32)
    Q = orc.synth_vectors(32 + dim, 0, nq, dim)
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            cap = 96   # bfloat16 sweep (32-bit rows of any dimension): 6 query blocks
            shared = nq - (nq % cap if nq % cap < 2 else 0)   # a tail below mq_min (2) gets its own sweep
            assert st["mq_queries"] == shared and st["mq_launches"] == (shared + cap - 1) // cap
        ix.set_option("multi_query", 0)
        ix.reset_stats()
        check(ix, rows, dim, Q[:9], 10)
        assert ix.stats()["mq_queries"] == 0


@pytest.mark.parametrize("bits", [4, 8, 16])
@pytest.mark.parametrize("dim,n", [(768, 2000), (384, 3000), (100, 4000), (37, 900), (3, 500)])
def test_shared_sweep_quantized_rows(bits, dim, n):
    """The MFMA sweep on 4/8/16-bit rows (decoded to exact integers in float32)."""
    rows = orc.synth_rows(131 + dim + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(132 + dim, 0, 40, dim)
    allow = np.arange(n) % 4 != 2
    with ScanIndex(dim, bits, SZG_COSINE) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=bits)
        assert ix.stats()["mq_queries"] == 40
        check(ix, rows, dim, Q[:20], 7, allow=allow, bits=bits)


def test_shared_sweep_two_shards():
    dim, n = 96, 6000
    rows = orc.synth_rows(55, 0, n, dim, 32)
    Q = orc.synth_vectors(56, 0, 48, dim)
    with ScanIndex(dim, 32, SZG_COSINE, devices=[0, 0]) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10)
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_launches"] == 2  # one shared sweep per shard


def test_shared_sweep_masks_tombstones_and_escalation():
    dim, n = 64, 4000
    rows = orc.synth_rows(77, 0, n, dim, 32)
    Q = orc.synth_vectors(78, 0, 24, dim)
    allow = np.arange(n) % 3 != 1
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 5, allow=allow)
        r, _, _ = ix.search_topk(Q, 5)
        dead = int(r[0, 0])
        ix.tombstone(dead)
        allow2 = np.ones(n, bool)
        allow2[dead] = False
        check(ix, rows, dim, Q, 5, allow=allow2)
        ix.set_option("force_escalate", 1)
        ix.reset_stats()
        check(ix, rows, dim, Q[:16], 5, allow=allow2)
        if DEFAULT_TUNABLES:
            assert ix.stats()["escalations"] == 16 and ix.stats()["mq_queries"] == 16
    if not TIE_EXACT:
        return
    # duplicates: ties across the candidate boundary -> escalation / exact replay
    base = orc.synth_vectors(5, 0, 6, 16)
    vecs = np.repeat(base, 200, axis=0)
    rows = orc.encode_rows(vecs, 32)
    Q = np.repeat(base[:4] + 0.01, 4, axis=0)
    with ScanIndex(16, 32, SZG_COSINE) as ix:
        ix.load(rows)
        check(ix, rows, 16, Q, 10)
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_queries"] == 16


def test_shared_sweep_zero_rows_and_zero_query():
    dim, n = 32, 600
    vecs = orc.synth_vectors(9, 0, n, dim)
    vecs[5] = 0.0
    vecs[77] = 0.0
    rows = orc.encode_rows(vecs, 32)
    Q = orc.synth_vectors(10, 0, 16, dim)
    Q[3] = 0.0
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 8)
        check(ix, rows, dim, Q, 600)  # k >= n: zero rows (distance exactly 1.0) included


@pytest.mark.parametrize("bits", [4, 8, 16, 32])
@pytest.mark.parametrize("dim,n", [(768, 2500), (384, 3000), (100, 4000), (3, 500)])
def test_shared_sweep_euclidean(bits, dim, n):
    """Euclidean metric through the MFMA sweep: key = |x|^2 - 2 x.q + |q|^2 in float32,
    certified with its own (looser) error bound; ids and float64 distances as the oracle's."""
    rows = orc.synth_rows(231 + dim + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(232 + dim, 0, 50, dim)
    Q[5] *= 30.0     # a query far outside the corpus
    Q[6] *= 1e-4     # and one at the origin
    allow = np.arange(n) % 5 != 1
    with ScanIndex(dim, bits, SZG_EUCLIDEAN) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=bits, metric=0)
        assert ix.stats()["mq_queries"] == 50
        check(ix, rows, dim, Q[:24], 100, allow=allow, bits=bits, metric=0)


def test_shared_sweep_euclidean_far_from_origin():
    """Rows clustered far from the origin: the expanded form loses digits to cancellation,
    the bound notices, and the queries escalate to the exact collect sweep -- same answers."""
    dim, n = 64, 5000
    rng = np.random.default_rng(9)
    vec = 50.0 + 1e-3 * rng.standard_normal((n, dim))
    rows = orc.encode_rows(vec, 32)
    Q = 50.0 + 1e-3 * rng.standard_normal((16, dim))
    with ScanIndex(dim, 32, SZG_EUCLIDEAN) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, metric=0)
        st = ix.stats()
        assert st["mq_queries"] == 16 and st["escalations"] > 0


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("dim,n", [(768, 3000), (1536, 1500), (100, 4000), (37, 900), (64, 2000), (128, 2500), (3, 500)])
def test_shared_sweep_int8_mfma(bits, metric, dim, n):
    """8- and 4-bit rows take the exact integer sweep (v_mfma_i32_16x16x64_i8 on the queries'
    int8 digit planes; 4-bit rows as two nibble operands per piece); same bar, and with the shared sweeps switched
    off (multi_query=0: one sweep per query) the answers are the same."""
    rows = orc.synth_rows(331 + dim, 0, n, dim, bits)
    Q = orc.synth_vectors(332 + dim, 0, 48, dim)
    Q[7] *= 25.0
    Q[8] *= 1e-3
    allow = np.arange(n) % 3 != 0
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        for i8 in (1, 0):
            ix.set_option("multi_query", i8)
            ix.reset_stats()
            check(ix, rows, dim, Q, 10, bits=bits, metric=metric)
            assert ix.stats()["mq_queries"] == (48 if i8 else 0)
            check(ix, rows, dim, Q[:17], 33, allow=allow, bits=bits, metric=metric)


@pytest.mark.parametrize("bits", [8, 32])
def test_fused_selection_overflow_falls_back(bits):
    """The shared sweep collects (query, row) pairs under a threshold taken from a prefix of the
    rows.  When the prefix is unrepresentative -- rows sorted worst-first, or one vector repeated
    thousands of times -- the candidate buffers overflow and the batch is redone through the
    score matrix; answers stay the oracle's either way (and equal with force_matrix=1)."""
    dim, n = 32, 60000
    rng = np.random.default_rng(3)
    q0 = rng.standard_normal(dim)
    vec = rng.uniform(-1, 1, (n, dim))
    order = np.argsort(vec @ q0)              # worst match first, best last
    vec = vec[order]
    vec[1000:31000] = vec[500]                # 30 000 copies of one vector
    rows = orc.encode_rows(vec, bits)
    Q = np.vstack([q0 + 0.01 * rng.standard_normal(dim) for _ in range(16)])
    Q[3] = vec[500]                           # the repeated vector itself
    with ScanIndex(dim, bits, SZG_COSINE) as ix:
        ix.load(rows)
        ix.set_option("tie_mode", 1)          # ties here are by construction; order among them is not the point
        r1, d1, c1 = ix.search_topk(Q, 10)
        assert ix.stats()["mq_fallbacks"] == 1
        ix.set_option("force_matrix", 1)
        r0, d0, c0 = ix.search_topk(Q, 10)
        assert ix.stats()["mq_queries"] == 32
        for qi in range(Q.shape[0]):
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, 1, Q[qi], k=10)
            for d in (d1, d0):
                got = np.sort(d[qi, : c1[qi]])
                want = np.sort(o_dist)
                assert ((got == want) | (np.isnan(got) & np.isnan(want))).all(), qi


# ---- bfloat16 shared sweep (32-bit rows, any dimension) --------------------------------------------

@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("dim,n", [(768, 3000), (48, 6000), (16, 9000), (80, 4000), (1040, 2500), (1536, 2000),
                                   (100, 4000), (37, 3000), (3, 2500), (33, 5000), (770, 2500), (8, 6000), (63, 3000),
                                   (1001, 2000)])
def test_bf16_sweep_matches_oracle(metric, dim, n):
    """Any dimension: whole 128-byte steps, a last step of 1..7 16-byte pieces (read past the row as zeros), a last
    piece with padding; two-stage selection; answers identical to the reference loop's."""
    rows = orc.synth_rows(900 + dim, 0, n, dim, 32)
    Q = orc.synth_vectors(901 + dim, 0, 50, dim)
    with ScanIndex(dim, 32, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, metric=metric)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            # 50 queries: one sweep of 4 query blocks where the image fits LDS (a KiB per 32 elements and block)
            blocks = min(6, (160 * 1024 - 13600) // (((dim + 31) // 32) * 1024))
            assert st["mq_bf16_sweeps"] == st["mq_launches"] == (50 + 16 * blocks - 1) // (16 * blocks)
            assert st["mq_fallbacks"] == 0
            ix.reset_stats()
            check(ix, rows, dim, Q[:30], 10, metric=metric)     # two query blocks: the NB = 2 kernels
            check(ix, rows, dim, Q[:9], 10, metric=metric)      # one
            assert ix.stats()["mq_bf16_sweeps"] == 2
        ix.set_option("multi_query", 0)
        ix.reset_stats()
        check(ix, rows, dim, Q[:20], 10, metric=metric)
        assert ix.stats()["mq_bf16_sweeps"] == 0


@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("dim,n", [(768, 3000), (384, 4000), (64, 6000), (8, 5000), (40, 4000), (72, 3000), (1000, 2500),
                                   (1536, 2000), (300, 3000), (100, 4000), (37, 3000), (3, 2500), (9, 4000), (65, 3000)])
def test_bf16_sweep_16bit_rows(metric, dim, n):
    """16-bit rows go through the bfloat16 sweep too (codes decoded to n = 2v - 65535 on the fly, two K-steps per
    128-byte step, short last steps, padding codes inside the last piece taken off the norm): answers identical to
    the reference loop's, with and without a filter."""
    rows = orc.synth_rows(1900 + dim, 0, n, dim, 16)
    Q = orc.synth_vectors(1901 + dim, 0, 50, dim)
    allow = np.arange(n) % 5 != 1
    with ScanIndex(dim, 16, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=16, metric=metric)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            assert st["mq_bf16_sweeps"] == st["mq_launches"] >= 1 and st["mq_queries"] == 50
            assert st["mq_fallbacks"] == 0
        check(ix, rows, dim, Q[:20], 7, allow=allow, bits=16, metric=metric)
        ix.set_option("multi_query", 0)
        ix.reset_stats()
        check(ix, rows, dim, Q[:20], 10, bits=16, metric=metric)
        assert ix.stats()["mq_bf16_sweeps"] == 0


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("dim", [1, 2, 3, 5, 16])
def test_bf16_sweep_few_dims_far_from_origin(dim, fused):
    """Every row and query within a few degrees of one direction, 1..16 dimensions: with so few elements their
    bfloat16 roundings can all point the same way, and the sweep's key is off by up to 2^-7 (two operands at 2^-8
    each) -- the certification band must be that wide (scripts/fuzz_gpu.py seed 311 found it at half)."""
    n = 5000
    for seed in range(6):
        rng = np.random.default_rng(7000 + 10 * dim + seed)
        vec = rng.uniform(-1, 1, (n, dim)) * 1e3 + 5e3
        Q = rng.uniform(-1, 1, (8, dim)) * 1e3 + 5e3
        rows = orc.encode_rows(vec, 32)
        with ScanIndex(dim, 32, SZG_COSINE) as ix:
            ix.load(rows)
            ix.set_option("force_matrix", 1 - fused)
            ix.set_option("mq_min", 8)
            check(ix, rows, dim, Q, 1)
            check(ix, rows, dim, Q, 10, allow=rng.random(n) < 0.5)
            if DEFAULT_TUNABLES:
                assert ix.stats()["mq_bf16_sweeps"] >= 2


def test_bf16_sweep_near_duplicates_and_scales():
    """Rows that bfloat16 cannot tell apart (relative differences of 1e-4 .. 1e-7, far below 2^-8), rows
    scaled by 1e+-18 (the same direction: equal cosine keys, the norms near the float32 range ends) and
    exact duplicates: the float32 re-score and the float64 re-rank decide, the order is the reference's."""
    rng = np.random.default_rng(77)
    dim, n = 256, 6000
    base = rng.standard_normal((8, dim))
    V = rng.standard_normal((n, dim))
    for i in range(400):          # clusters of near-copies of 8 directions
        V[i] = base[i % 8] * (1.0 + rng.standard_normal(dim) * 10.0 ** -(4 + i % 4))
    V[400:420] = base[0] * 1e18
    V[420:440] = base[1] * 1e-18
    V[440:460] = base[2]          # exact duplicates: equal distances -> the exact replay answers
    rows = orc.encode_rows(V, 32)
    Q = np.concatenate([base + rng.standard_normal((8, dim)) * 1e-3, rng.standard_normal((24, dim))])
    for metric in (SZG_COSINE, SZG_EUCLIDEAN):
        with ScanIndex(dim, 32, metric) as ix:
            ix.load(rows)
            check(ix, rows, dim, Q, 25, metric=metric)
            if DEFAULT_TUNABLES:
                assert ix.stats()["mq_bf16_sweeps"] >= 1


def test_bf16_sweep_masks_and_k_beyond_the_slack():
    """Filter masks and tombstones act in the bfloat16 sweep's hit path; k = 200 (kp = 300)."""
    dim, n = 128, 20000
    rows = orc.synth_rows(950, 0, n, dim, 32)
    Q = orc.synth_vectors(951, 0, 33, dim)
    allow = np.arange(n) % 3 != 1
    with ScanIndex(dim, 32, SZG_COSINE) as ix:
        ix.load(rows)
        for r in range(0, n, 7):
            ix.tombstone(r)
        live = np.ones(n, bool)
        live[::7] = False
        r, d, c = ix.search_topk(Q, 200, allow=np.tile(allow, (33, 1)))
        for qi in range(33):
            o_rows, o_dist, _ = orc.search_exact(rows, dim, 32, SZG_COSINE, Q[qi], k=200,
                                                 allow=(allow & live).astype(np.uint8))
            assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows]
            assert (d[qi, : c[qi]] == o_dist).all()
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_bf16_sweeps"] == 1


@pytest.mark.parametrize("bits", [8, 4])
@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
def test_int8_sweep_two_query_groups_per_launch(bits, metric):
    """More than 48 queries on 8- / 4-bit rows: one launch walks the passes of two groups of 48 back to back
    (hits, thresholds and score-matrix rows indexed by 48 * group + query); answers unchanged."""
    dim, n = 128, 6000
    rows = orc.synth_rows(990 + bits, 0, n, dim, bits)
    Q = orc.synth_vectors(991, 0, 70, dim)
    allow = np.arange(n) % 5 != 3
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=bits, metric=metric)
        check(ix, rows, dim, Q, 10, allow=allow, bits=bits, metric=metric)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            # 2 calls x (48 + 22); 8-bit rows in whole 64-byte steps take the bfloat16 sweep, 70 queries in ONE pass
            assert st["mq_queries"] == 140 and st["mq_launches"] == (2 if st["mq_bf16_sweeps"] else 4)
        ix.set_option("force_matrix", 1)                                # the score-matrix form
        check(ix, rows, dim, Q, 10, bits=bits, metric=metric)
        ix.set_option("force_matrix", 0)
        check(ix, rows, dim, Q[:60], 10, bits=bits, metric=metric)


# ---- long calls: the finished batches are assembled on a second host thread ----------------------------------------

@pytest.mark.parametrize("bits,dim", [(8, 64), (4, 128), (16, 64), (32, 48), (32, 20)])
@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
def test_long_calls_with_and_without_the_finisher_thread(bits, dim, metric):
    """Calls of 3+ shared-sweep batches (here 330 and 530 queries) hand their finished batches to a second host
    thread; the answers are the oracle's, with a filter, with forced escalations, and equal to the one-thread form's."""
    n = 2500
    rows = orc.synth_rows(4100 + bits + dim, 0, n, dim, bits)
    Q = orc.synth_vectors(4101 + bits + dim, 0, 530, dim)
    rng = np.random.default_rng(bits * 1000 + dim)
    allow = rng.random((530, n)) < 0.6
    with ScanIndex(dim, bits, metric) as ix:
        ix.load(rows)
        r1, d1, c1 = ix.search_topk(Q, 10)
        r1m, d1m, c1m = ix.search_topk(Q[:330], 7, allow=allow[:330])
        ix.set_option("force_escalate", 1)
        r1e, d1e, c1e = ix.search_topk(Q[:330], 3)
        ix.set_option("force_escalate", 0)
        ix.set_option("finish_thread", 0)
        r0, d0, c0 = ix.search_topk(Q, 10)
        r0m, d0m, c0m = ix.search_topk(Q[:330], 7, allow=allow[:330])
        assert (r1 == r0).all() and (c1 == c0).all() and ((d1 == d0) | (np.isnan(d1) & np.isnan(d0))).all()
        assert (r1m == r0m).all() and (c1m == c0m).all() and ((d1m == d0m) | (np.isnan(d1m) & np.isnan(d0m))).all()
    for qi in list(range(0, 530, 37)) + [329, 529]:
        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=10)
        assert [int(x) for x in r1[qi, : c1[qi]]] == [int(x) for x in o_rows], qi
        assert (d1[qi, : c1[qi]] == np.asarray(o_dist)).all(), qi
        if qi < 330:
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=7, allow=allow[qi].astype(np.uint8))
            assert [int(x) for x in r1m[qi, : c1m[qi]]] == [int(x) for x in o_rows], qi
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=3)
            assert [int(x) for x in r1e[qi, : c1e[qi]]] == [int(x) for x in o_rows], qi
            assert (d1e[qi, : c1e[qi]] == np.asarray(o_dist)).all(), qi


# ---- bfloat16 shared sweep on 64-bit rows: the reference's DEFAULT quantization (collection.go:254-256) -------------

@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("dim,n", [(768, 2500), (384, 3000), (16, 6000), (17, 5000), (31, 4000), (48, 4000), (100, 3000),
                                   (1000, 1500), (3, 2500), (1, 900), (33, 3000), (2, 2000)])
def test_bf16_sweep_64bit_rows(metric, dim, n):
    """64-bit rows (quantization.go:8-9, 29-30: the float64 as it is) through the shared sweep: two float64 per
    16-byte chunk narrowed to float32 and rounded to bfloat16 on the fly, a 128-byte step of a row being HALF a
    K-step of the matrix instruction (odd step counts, short last steps, an odd dimension's padding element);
    candidates re-scored in float32, re-ranked in float64: ids and distances are the reference loop's."""
    rows = orc.synth_rows(6400 + dim, 0, n, dim, 64)
    Q = orc.synth_vectors(6401 + dim, 0, 50, dim)
    allow = np.arange(n) % 5 != 1
    with ScanIndex(dim, 64, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=64, metric=metric)
        st = ix.stats()
        if DEFAULT_TUNABLES:
            assert st["mq_queries"] == 50 and st["mq_bf16_sweeps"] == st["mq_launches"] >= 1
        check(ix, rows, dim, Q[:20], 7, allow=allow, bits=64, metric=metric)
        ix.set_option("force_matrix", 1)      # the score-matrix form (lists of bfloat16 keys, wide slack)
        check(ix, rows, dim, Q[:20], 10, bits=64, metric=metric)
        ix.set_option("force_matrix", 0)
        ix.set_option("multi_query", 0)
        ix.reset_stats()
        check(ix, rows, dim, Q[:9], 10, bits=64, metric=metric)
        assert ix.stats()["mq_queries"] == 0


@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
def test_bf16_sweep_64bit_rows_beyond_float32(metric):
    """float64 rows the sweep's float32 cannot hold: elements of 1e200 (the narrowed value is inf), 1e-200 (it is 0),
    rows that differ only beyond float32's 24 bits, NaN and Inf elements, zero rows.  The sweep forces what it cannot
    rank into the candidates; the float64 re-rank and the reference's selection decide."""
    dim, n = 40, 4000
    rng = np.random.default_rng(64)
    vec = rng.uniform(-1, 1, (n, dim))
    vec[5] *= 1e200
    vec[6] *= 1e-200
    vec[7] = 0.0
    vec[100:140] = vec[99] * (1.0 + np.arange(1, 41)[:, None] * 1e-12)   # equal in float32, not in float64
    vec[200, 3] = np.nan
    vec[201, 4] = np.inf
    vec[2000] = vec[99] * 1e200
    rows = orc.encode_rows(vec, 64)
    Q = np.vstack([vec[99] + 1e-3 * rng.standard_normal(dim) for _ in range(6)] +
                  [rng.uniform(-1, 1, dim) for _ in range(14)] + [vec[6] * 1e200, vec[5] * 1e-200])
    with ScanIndex(dim, 64, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 12, bits=64, metric=metric)
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_queries"] == len(Q)


@pytest.mark.parametrize("bits", [4, 8, 16])
@pytest.mark.parametrize("dim", [768, 384, 100])
def test_resident_row_norms_follow_mutations(bits, dim):
    """The shared sweeps of 4-, 8- and 16-bit rows take the rows' norms from a resident array (built on the device
    before the first batch, caught up after appends, refreshed at once when a row is overwritten -- through either
    entry point): batches after every kind of mutation answer as the reference's loop over the mutated corpus."""
    n = 2500
    rows = orc.synth_rows(7100 + dim + bits, 0, n, dim, bits).reshape(n, -1).copy()
    Q = orc.synth_vectors(7101 + dim, 0, 48, dim)
    with ScanIndex(dim, bits, SZG_COSINE) as ix:
        ix.load(rows[:2000])
        check(ix, rows[:2000].reshape(-1), dim, Q, 10, bits=bits)           # norms built for 2 000 rows
        ix.append(rows[2000:2300])                                           # appended rows: caught up
        check(ix, rows[:2300].reshape(-1), dim, Q, 10, bits=bits)
        # overwrite stored bytes: the rows the queries like best get the bytes of other rows
        r, _, _ = ix.search_topk(Q[:1], 3)
        for j, victim in enumerate(int(x) for x in r[0, :3]):
            rows[victim] = rows[2400 + j]
            ix.overwrite(victim, rows[victim])
        check(ix, rows[:2300].reshape(-1), dim, Q, 10, bits=bits)
        # overwrite from a float64 vector (quantized on the device): the mirror's bytes are the oracle's quantization
        v = orc.synth_vectors(7102 + dim, 0, 1, dim)[0] * 0.25
        victim = int(ix.search_topk(Q[1:2], 1)[0][0, 0])
        ix.overwrite_vector(victim, v)
        rows[victim] = ix.read_rows(victim, 1)[0]
        check(ix, rows[:2300].reshape(-1), dim, Q, 10, bits=bits)
        ix.tombstone(int(r[0, 0]))
        ix.append(rows[2300:2500])
        allow = np.ones(2500, dtype=bool)
        allow[int(r[0, 0])] = False
        got_r, got_d, got_c = ix.search_topk(Q, 10)
        for qi in range(Q.shape[0]):
            o_rows, o_dist, _ = orc.search_exact(rows.reshape(-1), dim, bits, SZG_COSINE, Q[qi], k=10, allow=allow.astype(np.uint8))
            assert [int(x) for x in got_r[qi, : got_c[qi]]] == [int(x) for x in o_rows], qi
            assert (got_d[qi, : got_c[qi]] == o_dist).all(), qi
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_queries"] > 0
        ix.load(rows[:1000])                                                 # a reload starts the norms over
        check(ix, rows[:1000].reshape(-1), dim, Q, 10, bits=bits)


@pytest.mark.parametrize("metric", [SZG_COSINE, SZG_EUCLIDEAN])
@pytest.mark.parametrize("dim,n", [(768, 3000), (384, 4000), (128, 9000), (64, 2000), (120, 2500), (500, 2000), (100, 3000),
                                   (37, 1500)])
def test_8bit_rows_through_the_bf16_sweep(metric, dim, n):
    """8-bit rows whose pitch is a whole number of 64-byte steps (the tiled layout) share ONE bfloat16 pass per 96
    queries: the codes are exact in bfloat16, only the query is rounded; the band is re-scored in float32 and everything
    re-ranked in float64 -- ids and distances are the reference loop's.  Other 8-bit shapes and every radius batch
    stay on the exact int8 sweep."""
    rows = orc.synth_rows(8800 + dim, 0, n, dim, 8)
    Q = orc.synth_vectors(8801 + dim, 0, 96, dim)
    allow = np.arange(n) % 6 != 1
    tiled = ((dim + 15) // 16 * 16) % 64 == 0   # the row pitch (16-byte pieces) is a whole number of 64-byte steps: 120 and 500 pad
    with ScanIndex(dim, 8, metric) as ix:
        ix.load(rows)
        check(ix, rows, dim, Q, 10, bits=8, metric=metric)
        st = ix.stats()
        if DEFAULT_TUNABLES and not os.environ.get("SZG_BF16_8BIT") and not os.environ.get("SZG_NO_ROW_NORMS"):
            assert (st["mq_bf16_sweeps"] > 0) == tiled
            assert st["mq_launches"] == (1 if tiled else 2)
        check(ix, rows, dim, Q[:40], 7, allow=allow, bits=8, metric=metric)
        check(ix, rows, dim, Q[:20], 100, bits=8, metric=metric)          # kp beyond the band's usual size
        ix.tombstone(5)
        ix.append(rows[:50].reshape(50, -1))                             # duplicates of the first rows: ties at the top
        rows2 = np.concatenate([rows.reshape(n, -1), rows.reshape(n, -1)[:50]])
        live = np.ones(n + 50, dtype=bool)
        live[5] = False
        r, d, c = ix.search_topk(Q[:30], 10)
        for qi in range(30):
            o_rows, o_dist, _ = orc.search_exact(rows2.reshape(-1), dim, 8, metric, Q[qi], k=10, allow=live.astype(np.uint8))
            assert [int(x) for x in r[qi, : c[qi]]] == [int(x) for x in o_rows], qi
            assert (d[qi, : c[qi]] == o_dist).all(), qi
        ix.reset_stats()
        radii = [float(d[qi, 5]) for qi in range(30)]
        hits = ix.search_radius_batch(Q[:30], radii)
        for qi in range(30):
            w_r, w_d, _ = orc.search_exact(rows2.reshape(-1), dim, 8, metric, Q[qi], radius=radii[qi], allow=live.astype(np.uint8))
            assert [int(x) for x in hits[qi][0]] == [int(x) for x in w_r] and (np.asarray(hits[qi][1]) == w_d).all(), qi
        if DEFAULT_TUNABLES:
            assert ix.stats()["mq_bf16_sweeps"] == 0                      # radius batches: the exact int8 sweep
