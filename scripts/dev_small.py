"""Wall QPS and kernel time of single-query sweeps on small shards (the 8-GPU regime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim, bits, metric, k = 768, 32, 1, 11
nq = 4096
q = synth_vectors(99, 0, nq, dim)
for n in [int(x) for x in (sys.argv[1:] or ["125056", "250048", "500032", "1000000"])]:
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.synth(n, 1234)
        ix.set_option('multi_query', 0)
        for name, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
            ix.set_option(name, int(val))
        ix.search_topk(q[:256], k)
        t0 = time.perf_counter()
        ix.search_topk(q, k)
        wall = time.perf_counter() - t0
        # chunks of 256 as the sharded searcher issues them
        t0 = time.perf_counter()
        for i in range(0, nq, 256):
            ix.search_topk(q[i:i + 256], k)
        wall256 = time.perf_counter() - t0
        ix.set_timing(True)
        ix.reset_stats()
        ix.search_topk(q[:1024], k)
        s = ix.stats()
        ms = s["scan_ms"] / max(s["scan_bytes"] / (n * 3072.0), 1)  # per sweep
        print("rows %8d  one call %.0f QPS  chunks of 256 %.0f QPS  scan %.1f us (%.2f TB/s)  esc %d" % (
            n, nq / wall, nq / wall256, ms * 1e3, n * 3072 / ms / 1e9, s["escalations"]), flush=True)
