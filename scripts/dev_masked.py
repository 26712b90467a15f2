"""Filtered single-query sweeps: wall per query and algorithmic TB/s at several pass rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
from syzgydb_amd.index import pack_allow_bits
n, dim, bits, metric, k = 1000000, 768, int(os.environ.get("SZG_BITS", "32")), 1, 10
q = synth_vectors(99, 0, 256, dim)
rng = np.random.default_rng(1)
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    rates = [float(x) for x in os.environ.get("SZG_RATES", "1.0,0.5,0.05,0.001").split(",")]
    for mqo in (0, 1):
        ix.set_option("multi_query", mqo)
        for rate in rates:
            mask = pack_allow_bits(rng.random(n) < rate)
            masks = np.tile(mask, (256, 1))
            ix.search_topk(q[:32], k, allow=masks[:32])
            t0 = time.perf_counter()
            ix.search_topk(q, k, allow=masks)
            el = time.perf_counter() - t0
            print("multi_query=%d pass rate %.3f: %.0f queries/s (%.3f ms per query; %.2f TB/s algorithmic)" % (
                mqo, rate, 256 / el, el / 256 * 1e3, 256 * n * ix.row_bytes / el / 1e12), flush=True)
