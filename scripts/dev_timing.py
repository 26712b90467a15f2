"""Does per-kernel event timing (szg_set_timing) cost throughput?  Same handle, timing off / on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim, bits, metric, k = 768, 32, 1, 10
nq = 2048
q = synth_vectors(99, 0, nq, dim)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.set_option('multi_query', 0)
    ix.search_topk(q[:512], k)
    for rep in range(2):
        for timing in (False, True):
            ix.set_timing(timing)
            ix.reset_stats()
            t0 = time.perf_counter()
            ix.search_topk(q, k)
            wall = time.perf_counter() - t0
            s = ix.stats()
            print("timing %-5s: %.0f q/s  enqueue %.2f us/query  scan %.1f us/sweep" % (
                timing, nq / wall, s["host_enqueue_us"] / nq,
                1e3 * s["scan_ms"] / max(s["timed_launches"], 1) / 16), flush=True)
