"""The oracle against the reference's own known answers (tests/golden/
reference_kats.json) and against the committed fixtures.  CPU only."""
import math

import numpy as np
import pytest

from golden_util import case_rows, hex_f64, hex_list, load_cases, load_kats, same_f64


def test_euclidean_kat(oracle):
    for kat in load_kats()["euclidean"]:
        assert oracle.euclidean(kat["a"], kat["b"]) == kat["expect"]  # exact, as the reference asserts


def test_exhaustive_search_kat(oracle):
    kat = load_kats()["exhaustive_search"]
    ids = sorted(int(i) for i in kat["docs"])
    # visit order of the seeded reference: sort.Strings over the decimal ids
    order = oracle.sorted_id_order(ids)
    vecs = np.array([kat["docs"][str(ids[int(i)])] for i in order])
    rows = oracle.encode_rows(vecs, 64)
    r, d, searched = oracle.search_exact(rows, 3, 64, oracle.EUCLIDEAN, kat["query"], k=kat["k"])
    got_ids = [ids[int(order[int(x)])] for x in r]
    assert got_ids == kat["expect_ids"]
    assert list(d) == kat["implied_distances"]
    assert searched / len(ids) * 100 == kat["expect_percent_searched"]


def test_angular_kats(oracle):
    kat = load_kats()["angular_libm"]
    for c in kat["cases"]:
        assert abs(oracle.angular(kat["a"], c["b"]) - c["expect"]) <= 1e-14
    assert oracle.angular(kat["a"], kat["a"]) == 0.0
    assert oracle.angular([0, 0, 0], [1, 2, 3]) == 1.0   # collection.go:828-830
    assert oracle.angular([1, 2, 3], [0, 0, 0]) == 1.0


def test_go_acos_tracks_libm(oracle):
    xs = np.linspace(-1, 1, 20001)
    err = max(abs(oracle.go_acos(x) - math.acos(x)) for x in xs)
    assert err < 4e-15
    assert math.isnan(oracle.go_acos(1.0000000000000002))  # no clamp, collection.go:831
    assert oracle.go_acos(1.0) == 0.0


def test_quantizer_kats(oracle):
    kats = load_kats()
    assert [oracle.dequantize(v, 4) for v in range(16)] == kats["dequantize_4bit_table"]["expect"]
    for c in kats["quantize"]["cases"]:
        assert oracle.quantize(c["value"], c["bits"]) == c["expect"]
    assert oracle.dequantize(128, 8) == kats["quantize"]["dequantize_128_8bit"]
    # math.Round is half away from zero: (v+1)/2*15 == 7.5 for v == 0 -> 8
    assert oracle.quantize(0.0, 4) == 8


def test_encode_kats(oracle):
    kat = load_kats()["encode_vector"]
    for bits, hx in kat["bytes_hex"].items():
        assert oracle.encode_vector(kat["vector"], int(bits)).tobytes().hex() == hx
    for bits, dim, size in load_kats()["vector_size"]["cases"]:
        assert oracle.vector_size(bits, dim) == size
    assert oracle.vector_size(7, 3) == -1  # the reference panics


def test_roundtrips(oracle):
    kat = load_kats()["roundtrip_exact"]
    v = np.array(kat["q64"])
    assert same_f64(oracle.decode_vector(oracle.encode_vector(v, 64), v.size, 64), v)
    v = np.array(kat["q32_integers"])
    assert same_f64(oracle.decode_vector(oracle.encode_vector(v, 32), v.size, 32), v)


def test_sorted_id_order(oracle):
    kat = load_kats()["sorted_id_order"]
    perm = oracle.sorted_id_order(kat["ids"])
    assert [kat["ids"][int(i)] for i in perm] == kat["expect_visit"]


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: "c%d-q%d-m%d-d%d" % (c["id"], c["bits"], c["metric"], c["dim"]))
def test_oracle_reproduces_fixtures(oracle, case):
    rows = case_rows(case, oracle)
    q = hex_list(case["query_hex"])
    dim, bits, metric = case["dim"], case["bits"], case["metric"]
    r, d, searched = oracle.search_exact(rows, dim, bits, metric, q, k=case["topk"]["k"])
    assert [int(x) for x in r] == case["topk"]["rows"] and same_f64(d, hex_list(case["topk"]["dist_hex"]))
    assert searched == case["n"]
    radius = hex_f64(case["radius"]["radius_hex"])
    r, d, _ = oracle.search_exact(rows, dim, bits, metric, q, radius=radius)
    assert [int(x) for x in r] == case["radius"]["rows"] and same_f64(d, hex_list(case["radius"]["dist_hex"]))
    assert (d <= radius).all()
    allow = (np.arange(case["n"]) % 3 != 0).astype(np.uint8)
    r, d, _ = oracle.search_exact(rows, dim, bits, metric, q, k=case["filtered"]["k"], allow=allow)
    assert [int(x) for x in r] == case["filtered"]["rows"] and same_f64(d, hex_list(case["filtered"]["dist_hex"]))


def test_search_semantics(oracle):
    """Edge cases of consider()."""
    dim, bits = 4, 64
    vecs = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0]], dtype=float)
    rows = oracle.encode_rows(vecs, bits)
    q = [1, 0, 0, 0]
    # radius wins over k (collection.go:598-606)
    r, d, _ = oracle.search_exact(rows, dim, bits, oracle.EUCLIDEAN, q, k=1, radius=0.5)
    assert sorted(int(x) for x in r) == [0, 2, 4]
    # strict '>' at the boundary: first visited wins (collection.go:608)
    r, d, _ = oracle.search_exact(rows, dim, bits, oracle.EUCLIDEAN, q, k=2)
    assert sorted(int(x) for x in r) == [0, 2]
    # k > n returns everything, ascending
    r, d, _ = oracle.search_exact(rows, dim, bits, oracle.EUCLIDEAN, q, k=50)
    assert len(r) == 5 and list(d) == sorted(d)
    # listing mode (k == 0 and radius == 0) computes no distances
    r, d, s = oracle.search_exact(rows, dim, bits, oracle.EUCLIDEAN, q, k=0, radius=0.0, capacity=8)
    assert len(r) == 0 and s == 0
    # empty collection
    r, d, s = oracle.search_exact(np.zeros((0, 32), np.uint8), dim, bits, oracle.EUCLIDEAN, q, k=3)
    assert len(r) == 0 and s == 0
    # pointsSearched counts filtered rows too (collection.go:589 precedes :592)
    r, d, s = oracle.search_exact(rows, dim, bits, oracle.EUCLIDEAN, q, k=5, allow=[0, 1, 0, 1, 0])
    assert sorted(int(x) for x in r) == [1, 3] and s == 5


def test_nan_distance_heap_behaviour(oracle):
    """acos argument is not clamped (collection.go:831): a NaN is accepted only while
    the heap is not full and never displaces anything afterwards."""
    x = np.array([0.1, 0.7, 0.3])
    found = None
    rng = np.random.default_rng(3)
    for _ in range(2000):   # find a vector whose self-cosine rounds above 1
        v = rng.uniform(-1, 1, 3)
        if math.isnan(oracle.angular(v, v)):
            found = v
            break
    if found is None:
        pytest.skip("no NaN self-distance found")
    vecs = np.stack([x, found, -found])
    rows = oracle.encode_rows(vecs, 64)
    r, d, _ = oracle.search_exact(rows, 3, 64, oracle.COSINE, found, k=3)
    # both +v (cos just above 1) and -v (cos just below -1) fall outside acos' domain
    assert len(r) == 3 and np.isnan(d).sum() == 2


def test_synth_is_counter_based(oracle):
    a = oracle.synth_vectors(77, 0, 10, 5)
    b = oracle.synth_vectors(77, 4, 3, 5)
    assert (a[4:7] == b).all()
    assert ((a >= -1) & (a < 1)).all()
    assert (oracle.synth_rows(77, 4, 3, 5, 8) == oracle.encode_rows(b, 8)).all()
