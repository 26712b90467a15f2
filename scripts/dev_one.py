"""Run one scan config for a few queries (profiling target): SZG_LAUNCHES (1) calls of nq <= 16 queries, each ONE
query-major scan launch of nq sweeps."""
import os, sys
os.environ.setdefault("SZG_NO_EARLY_TAIL", "1")   # a short call as ONE launch (not 12 + 4 sweeps with an early tail)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits, metric, k, nq = [int(x) for x in sys.argv[1:7]]
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.set_option('multi_query', int(os.environ.get('SZG_MQ', '0')))
    q = synth_vectors(99, 0, nq, dim)
    if os.environ.get('SZG_RADIUS_HITS'):    # radius (collect) sweeps: radius = distance of the N-th neighbour of q[0]
        _, dd, _ = ix.search_topk(q[0], int(os.environ['SZG_RADIUS_HITS']))
        hits = ix.search_radius_batch(q, float(dd[0, -1]))
        print("done", sum(len(r) for r, _ in hits) / float(nq), "hits per query")
    else:
        for _ in range(int(os.environ.get('SZG_LAUNCHES', '1'))):
            r, d, c = ix.search_topk(q, k)
        print("done", r[0][:3])
