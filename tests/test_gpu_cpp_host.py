"""Builds tests/cpp/test_collection.cpp against include/syzgy_collection.hpp and the
C-ABI library and runs it: the reference's search tests through the C++ host mirror."""
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
import spanfile_writer as sw
from syzgydb_amd import codec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_collection_mirror(tmp_path):
    exe = tmp_path / "test_collection"
    lib_dir = os.path.join(ROOT, "syzgydb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_collection.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-lsyzgy_scan", "-Wl,-rpath," + lib_dir])
    # a collection file for NewCollection to open; expected answer from the oracle
    dim, bits, metric, n = 12, 8, 1, 300
    vecs = orc.synth_vectors(61, 0, n, dim)
    rows = codec.encode_rows(vecs, bits)
    ids = list(range(100, 100 + n))
    path = tmp_path / "cpp.dat"
    sw.collection_file(path, metric, dim, bits, [(ids[i], b"m", rows[i].tobytes()) for i in range(n)])
    order = orc.sorted_id_order(ids)
    q = np.full(dim, 0.25)
    o_rows, o_dist, _ = orc.search_exact(rows[[int(i) for i in order]], dim, bits, metric, q, k=3)
    best = ids[int(order[int(o_rows[0])])]
    p = subprocess.run([str(exe), str(path), str(best), repr(float(o_dist[0]))], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "CPP_HOST_OK" in p.stdout
