"""Throughput of T threads issuing single-query searches on one handle (coalescing on/off)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits, metric, k = 1000000, 768, int(os.environ.get("SZG_BITS", "32")), 1, 10
q = synth_vectors(99, 0, 4096, dim)
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.search_topk(q[:64], k)
    for co in (0, 1):
        ix.set_option("coalesce", co)
        for T in (1, 4, 16, 64):
            per = 2048 // T if co else max(8, 512 // T)
            def worker(t):
                for i in range(per):
                    ix.search_topk(q[(t * per + i) % 4096], k)
            th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            [x.start() for x in th]; [x.join() for x in th]
            el = time.perf_counter() - t0
            print("coalesce=%d threads=%3d: %.0f queries/s  (%.2f ms per call)" % (co, T, T * per / el, el / per * 1e3), flush=True)
