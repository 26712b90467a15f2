"""Host cost per query of szg_search_topk (prepare / enqueue / assemble) and wall queries/s on
small shards -- the 8-GPU regime of the headline -- for one handle with 1..8 device shards."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim, bits, metric, k = 768, 32, 1, 11
nq = 4096
q = synth_vectors(99, 0, nq, dim)
for spec in (sys.argv[1:] or ["125056:1", "1000000:1", "1000000:8", "250048:2"]):
    n, nd = [int(x) for x in spec.split(":")]
    with ScanIndex(dim, bits, metric, devices=[0] * nd) as ix:
        ix.synth(n, 1234)
        ix.set_option('multi_query', 0)
        for name, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
            ix.set_option(name, int(val))
        ix.search_topk(q[:512], k)
        ix.reset_stats()
        t0 = time.perf_counter()
        for i in range(0, nq, 256):
            ix.search_topk(q[i:i + 256], k)
        wall = time.perf_counter() - t0
        s = ix.stats()
        print("rows %8d shards %d: %.0f q/s (%.1f us/query wall)  host us/query: prepare %.2f enqueue %.2f assemble %.2f" % (
            n, nd, nq / wall, 1e6 * wall / nq, s["host_prep_us"] / nq, s["host_enqueue_us"] / nq,
            s["host_finish_us"] / nq), flush=True)
