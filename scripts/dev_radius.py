"""cfg5-shard radius sweeps: kernel rate by number of concurrent callers (and the top-k sweep beside it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits = int(os.environ.get('SZG_ROWS', '12500000')), int(os.environ.get('SZG_DIM', '384')), int(os.environ.get('SZG_BITS', '4'))
q = synth_vectors(77, 0, 48, dim)
with ScanIndex(dim, bits, 1, devices=[0]) as ix:
    ix.synth(n, 4321)
    ix.set_option("multi_query", 0)
    for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(o, int(val))
    r, d, c = ix.search_topk(q[0], 500)
    radius = float(d[0][c[0] - 1])
    def report(tag, st, wall, nq):
        ms = st["scan_ms"] / max(st["timed_launches"], 1)
        per = st["scan_bytes"] / max(st["scan_launches"], 1)
        print("%-22s %6.0f q/s  launch %.3f ms  %.0f GB/s  (%d launches)" % (tag, nq / wall, ms, per / ms / 1e6, st["timed_launches"]), flush=True)
    for threads in (1, 2, 3, 4):
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(lambda v: ix.search_radius(v, radius), q[:8]))
            ix.set_timing(True); ix.reset_stats(); t0 = time.perf_counter()
            hits = list(ex.map(lambda v: len(ix.search_radius(v, radius)[0]), q))
            wall = time.perf_counter() - t0
        report("radius, %d callers" % threads, ix.stats(), wall, len(q))
    ix.search_topk(q[:16], 10); ix.reset_stats(); t0 = time.perf_counter(); ix.search_topk(q, 10); wall = time.perf_counter() - t0
    report("top-k 10 (16/launch)", ix.stats(), wall, len(q))
    ix.set_option("queries_per_launch", 1)
    ix.search_topk(q[:16], 10); ix.reset_stats(); t0 = time.perf_counter(); ix.search_topk(q, 10); wall = time.perf_counter() - t0
    report("top-k 10 (1/launch)", ix.stats(), wall, len(q))
    print("hits per query ~", sum(hits) / len(hits))
