"""The HIP path against the committed golden fixtures (tests/golden/)."""
import numpy as np
import pytest

import oracle as orc
from golden_util import case_rows, hex_f64, hex_list, load_cases, load_kats, same_f64
from syzgydb_amd import ScanIndex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: "c%d-q%d-m%d-d%d" % (c["id"], c["bits"], c["metric"], c["dim"]))
def test_hip_reproduces_fixtures(case):
    rows = case_rows(case, orc)
    q = hex_list(case["query_hex"])
    with ScanIndex(case["dim"], case["bits"], case["metric"]) as ix:
        ix.load(rows)
        k = case["topk"]["k"]
        r, d, c = ix.search_topk(q, k)
        assert [int(x) for x in r[0, : c[0]]] == case["topk"]["rows"]
        assert same_f64(d[0, : c[0]], hex_list(case["topk"]["dist_hex"]))
        radius = hex_f64(case["radius"]["radius_hex"])
        rr, dd = ix.search_radius(q, radius)
        assert [int(x) for x in rr] == case["radius"]["rows"]
        assert same_f64(dd, hex_list(case["radius"]["dist_hex"]))
        allow = np.arange(case["n"]) % 3 != 0
        k = case["filtered"]["k"]
        r, d, c = ix.search_topk(q, k, allow=allow)
        assert [int(x) for x in r[0, : c[0]]] == case["filtered"]["rows"]
        assert same_f64(d[0, : c[0]], hex_list(case["filtered"]["dist_hex"]))


def test_reference_kats_through_hip():
    kats = load_kats()
    kat = kats["exhaustive_search"]
    ids = sorted(int(i) for i in kat["docs"])
    vecs = np.array([kat["docs"][str(i)] for i in ids])
    with ScanIndex(3, 64, 0) as ix:
        ix.load(orc.encode_rows(vecs, 64))
        r, d, c = ix.search_topk(kat["query"], kat["k"])
        assert [ids[int(x)] for x in r[0]] == kat["expect_ids"]
        assert list(d[0]) == kat["implied_distances"]
    e = kats["euclidean"][0]
    with ScanIndex(3, 64, 0) as ix:
        ix.load(orc.encode_rows(np.array([e["b"]]), 64))
        r, d, c = ix.search_topk(e["a"], 1)
        assert d[0, 0] == e["expect"]   # collection_test.go:12-21, exact ==
