"""Dev probe: radius search latency/throughput on a cfg5-like shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits = 12_500_000, 384, 4
with ScanIndex(dim, bits, 1, devices=[0]) as ix:
    ix.synth(n, 1234)
    q = synth_vectors(99, 0, 40, dim)
    ix.set_option('multi_query', 0)
    r, d, c = ix.search_topk(q[:8], 100)
    print("top-100 distance range", d[0, 0], d[0, 99])
    for radius in (float(d[0, 99]), 0.42, 0.44):
        rr, dd = ix.search_radius(q[0], radius)
        ix.set_timing(True); ix.reset_stats()
        t0 = time.time()
        hits = 0
        for i in range(8, 24):
            rr, dd = ix.search_radius(q[i], radius)
            hits += len(rr)
        dt = time.time() - t0
        st = ix.stats()
        print("radius %.4f: %.1f hits/query, %.2f ms/query wall, scan %.3f ms (%.2f TB/s), launches %d" % (
            radius, hits / 16, dt / 16 * 1e3, st['scan_ms'] / st['timed_launches'],
            ix.rows * ix.row_bytes / (st['scan_ms'] / st['timed_launches']) / 1e9, st['scan_launches']))
        ix.set_timing(False)
