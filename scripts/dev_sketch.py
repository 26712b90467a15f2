"""Sketch pre-pass throughput: SZG_ROWS x SZG_DIM float32 cosine, one query per sweep and batched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, k = int(os.environ.get('SZG_ROWS', '1000000')), int(os.environ.get('SZG_DIM', '768')), 10
nq = int(os.environ.get('SZG_NQ', '1024'))
q = synth_vectors(99, 0, nq, dim)
metric = int(os.environ.get('SZG_METRIC', '1'))
k = int(os.environ.get('SZG_K', '10'))
with ScanIndex(dim, 32, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(o, int(val))
    base = None
    for sketch, multi, extra in ((0, 0, 30), (1, 0, 30)):
        if True:
            ix.set_option("sketch", sketch); ix.set_option("multi_query", multi); ix.set_option("sketch_extra", extra)
            ix.search_topk(q[:128], k)
            ix.set_timing(True); ix.reset_stats()
            t0 = time.perf_counter(); r, d, c = ix.search_topk(q, k); wall = time.perf_counter() - t0
            s = ix.stats()
            if base is None: base = (r.copy(), d.copy())
            same = bool((r == base[0]).all() and (d == base[1]).all())
            print("sketch=%d multi=%d extra=%d: %.0f QPS  scan %.3f ms/launch x %d  bytes/query %.0f MB  sketch_q %d fb %d esc %d  same %s" % (
                sketch, multi, extra, nq / wall, s["scan_ms"] / max(s["timed_launches"], 1), s["timed_launches"],
                s["scan_bytes"] / nq / 1e6, s["sketch_queries"], s["sketch_fallbacks"], s["escalations"], same), flush=True)
    # lone queries, one call each, one after the other (a single goroutine calling Search)
    for sketch in (0, 1):
        ix.set_option("sketch", sketch); ix.set_option("multi_query", 1); ix.set_timing(False)
        for i in range(20): ix.search_topk(q[i % nq], k)
        t0 = time.perf_counter()
        for i in range(200): ix.search_topk(q[i % nq], k)
        el = time.perf_counter() - t0
        print("sketch=%d lone queries: %.3f ms per call (%.0f calls/s)" % (sketch, el / 200 * 1e3, 200 / el), flush=True)
