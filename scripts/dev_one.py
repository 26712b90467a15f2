"""Run one scan config for a few queries (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits, metric, k, nq = [int(x) for x in sys.argv[1:7]]
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.set_option('multi_query', int(os.environ.get('SZG_MQ', '0')))
    ix.set_option('first_batch', 0)          # ONE query-major launch for the nq sweeps
    q = synth_vectors(99, 0, nq, dim)
    if os.environ.get('SZG_RADIUS_HITS'):    # radius (collect) sweeps: radius = distance of the N-th neighbour of q[0]
        _, dd, _ = ix.search_topk(q[0], int(os.environ['SZG_RADIUS_HITS']))
        hits = ix.search_radius_batch(q, float(dd[0, -1]))
        print("done", sum(len(r) for r, _ in hits) / float(nq), "hits per query")
    else:
        r, d, c = ix.search_topk(q, k)
        print("done", r[0][:3])
