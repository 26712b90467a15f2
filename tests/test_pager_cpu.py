"""The C++ spanfile pager (include/syzgy_pager.h) on files made by the test-side
writer: scan rules of spanfile.go:282-357.  Pure host code -> runs without a GPU."""
import numpy as np
import pytest

import spanfile_writer as sw
from syzgydb_amd import SpanfilePager, SzgError, codec


def test_writer_matches_survey_examples():
    # empty initial span, 15 bytes
    assert sw.span(0, "", []).hex() == "5350414e0000000f0000009e525b4c"
    # record id 42, seq 2, metadata {"a":"b"}, Q=32 vector [0.3, -0.5], 38 bytes
    vec = codec.encode_rows([[0.3, -0.5]], 32).tobytes()
    got = sw.span(2, "42", [(0, b'{"a":"b"}'), (1, vec)])
    assert got.hex() == ("5350414e0000002602023432020009" "7b2261223a2262227d" "0108"
                         "3e99999abf000000" "849bb4b5")
    # 7-code quirks (spanfile.go:569-575)
    assert sw.write7(126).hex() == "7e" and sw.write7(127).hex() == "807f"
    assert sw.write7(128).hex() == "8100" and sw.write7(3072).hex() == "9800"
    assert sw.write7(16382).hex() == "ff7e" and sw.write7(16383).hex() == "80ff7f"


def test_pager_reads_records_in_visit_order(tmp_path, oracle):
    dim, bits = 5, 8
    ids = [1, 2, 10, 100, 20, 3]
    vecs = oracle.synth_vectors(3, 0, len(ids), dim)
    rows = codec.encode_rows(vecs, bits)
    docs = [(i, b"meta-%d" % i, rows[n].tobytes()) for n, i in enumerate(ids)]
    path = tmp_path / "c.dat"
    sw.collection_file(path, 1, dim, bits, docs)
    with SpanfilePager(path) as pg:
        assert (pg.dim, pg.quant_bits, pg.metric) == (dim, bits, 1)
        assert pg.count == 6 and pg.skipped == 0
        got_ids = [int(x) for x in pg.ids()]
        assert got_ids == [1, 10, 100, 2, 20, 3]          # sort.Strings over decimal ids
        order = [ids.index(i) for i in got_ids]
        assert (pg.vectors() == rows[order]).all()
        assert [pg.metadata(r) for r in range(6)] == [b"meta-%d" % i for i in got_ids]


def test_pager_scan_rules(tmp_path, oracle):
    dim, bits = 4, 32
    v = codec.encode_rows(oracle.synth_vectors(5, 0, 6, dim), bits)
    docs = [(7, b"old", v[0].tobytes()), (8, b"eight", v[1].tobytes())]
    extra = [
        sw.span(50, "7", [(0, b"new"), (1, v[2].tobytes())]),                 # higher sequence wins
        sw.span(3, "8", [(0, b"stale"), (1, v[3].tobytes())]),                # lower sequence loses
        sw.span(60, "9", [(0, b"bad"), (1, v[4].tobytes())], corrupt=True),   # checksum fails: skipped
        sw.span(61, "11", [(0, b"gone"), (1, v[4].tobytes())], magic=sw.FREE),  # removed record
        sw.span(62, "12", [(0, b"pad"), (1, v[5].tobytes())], pad=9),         # padding before the CRC
        sw.span(63, "abc", [(0, b"x"), (1, v[5].tobytes())]),                 # non-numeric id: ignored by Search
        sw.span(64, "13", [(0, b"no vector")]),                               # no vector stream
    ]
    path = tmp_path / "rules.dat"
    sw.collection_file(path, 0, dim, bits, docs, extra_spans=extra)
    with SpanfilePager(path, n_threads=3) as pg:
        assert [int(x) for x in pg.ids()] == [12, 7, 8]
        assert pg.skipped == 1
        vec = pg.vectors()
        assert (vec[0] == v[5]).all() and (vec[1] == v[2]).all() and (vec[2] == v[1]).all()
        assert pg.metadata(1) == b"new" and pg.metadata(2) == b"eight" and pg.metadata(0) == b"pad"


def test_pager_large_file_parallel_crc(tmp_path, oracle):
    dim, bits, n = 16, 4, 5000
    rows = oracle.synth_rows(8, 0, n, dim, bits)
    docs = [(i, b"m", rows[i].tobytes()) for i in range(n)]
    path = tmp_path / "big.dat"
    sw.collection_file(path, 1, dim, bits, docs)
    with SpanfilePager(path, n_threads=4) as pg:
        assert pg.count == n and pg.skipped == 0
        ids = pg.ids()
        assert sorted(int(x) for x in ids) == list(range(n))
        assert [int(x) for x in ids] == sorted(range(n), key=str)
        assert (pg.vectors() == rows[[int(x) for x in ids]]).all()


def test_pager_errors(tmp_path):
    with pytest.raises(SzgError):
        SpanfilePager(tmp_path / "missing.dat")
    p = tmp_path / "nohdr.dat"
    p.write_bytes(sw.span(1, "5", [(0, b"m"), (1, b"\x00" * 8)]) + b"\x00" * 64)
    with pytest.raises(SzgError):
        SpanfilePager(p)


def test_oracle_span_builder_is_read_by_the_pager(tmp_path, oracle):
    """The spans the faithful CPU baseline walks are real spanfile spans: prepend a
    header record and the C++ pager reads every vector back."""
    import ctypes
    dim, bits, n = 10, 16, 200
    rows = oracle.synth_rows(21, 0, n, dim, bits)
    rb = rows.shape[1]
    spans = np.zeros(n * (rb + 80), dtype=np.uint8)
    offs = np.zeros(n, dtype=np.uint64)
    used = oracle.lib().orc_spans_build(rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n, dim, bits, 7,
                                        spans.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), spans.size,
                                        offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    assert used > 0
    path = tmp_path / "orc.dat"
    path.write_bytes(sw.header_span(1, "orc", 0, dim, bits) + spans[:used].tobytes() + b"\x00" * 64)
    with SpanfilePager(path) as pg:
        assert pg.count == n and pg.skipped == 0
        ids = [int(x) for x in pg.ids()]
        assert (pg.vectors() == rows[ids]).all()
        assert pg.metadata(0) == b"m" * 7
    s1, r1 = oracle.bench_topk(rows, dim, bits, 0, oracle.synth_vectors(22, 0, 2, dim), 5, 1)
    s2, r2 = oracle.bench_topk_faithful(rows, dim, bits, 0, oracle.synth_vectors(22, 0, 2, dim), 5, meta_len=7)
    assert (r1 == r2).all()


def test_pager_randomized_against_a_python_model(tmp_path, oracle):
    """Random collection files -- rewrites of the same id (highest sequence wins), removed
    records (FREE spans), checksum failures, padded spans, ids that sort differently as
    strings -- against a small Python model of the reference's scan rules
    (spanfile.go:282-357) and visit order (sort.Strings, :540-560)."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        dim = int(rng.choice([1, 3, 8, 33]))
        bits = int(rng.choice([4, 8, 16, 32, 64]))
        n_spans = int(rng.integers(0, 60))
        model = {}      # id -> (sequence, metadata, vector bytes)
        bad = 0
        spans = []
        for s in range(n_spans):
            seq = int(rng.integers(2, 5000))
            rid = int(rng.choice([1, 2, 3, 10, 11, 21, 100, 101, 9, 1000, int(rng.integers(0, 10**9))]))
            vec = codec.encode_rows(rng.uniform(-1, 1, (1, dim)), bits)[0].tobytes()
            meta = bytes(rng.integers(0, 256, int(rng.integers(0, 20))).astype(np.uint8))
            kind = rng.random()
            if kind < 0.12:
                spans.append(sw.span(seq, str(rid), [(0, meta), (1, vec)], corrupt=True))
                bad += 1
            elif kind < 0.24:
                spans.append(sw.span(seq, str(rid), [(0, meta), (1, vec)], magic=sw.FREE))
            else:
                spans.append(sw.span(seq, str(rid), [(0, meta), (1, vec)], pad=int(rng.choice([0, 0, 3, 14]))))
                if rid not in model or seq > model[rid][0]:
                    model[rid] = (seq, meta, vec)
        path = tmp_path / ("rand%d.dat" % case)
        sw.collection_file(path, int(rng.integers(0, 2)), dim, bits, [], extra_spans=spans,
                           tail_zeros=int(rng.choice([0, 15, 4096])))
        order = sorted(model, key=lambda i: str(i))
        with SpanfilePager(path, n_threads=int(rng.integers(1, 5))) as pg:
            assert [int(x) for x in pg.ids()] == order, case
            assert pg.skipped == bad
            vec = pg.vectors()
            for row, rid in enumerate(order):
                assert vec[row].tobytes() == model[rid][2]
                assert pg.metadata(row) == model[rid][1]
