"""Same-index A/B of one option: SZG_AB=name:v1,v2[,..]  rows...  (alternates twice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim = int(os.environ.get('SZG_DIM', '768')); bits = int(os.environ.get('SZG_BITS', '32'))
metric = int(os.environ.get('SZG_METRIC', '1')); k = int(os.environ.get('SZG_K', '11'))
nq = int(os.environ.get('SZG_NQ', '4096'))
name, vals = os.environ['SZG_AB'].split(':')
vals = [int(v) for v in vals.split(',')]
q = synth_vectors(99, 0, nq, dim)
for n in [int(x) for x in sys.argv[1:]]:
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.synth(n, 1234)
        ix.set_option('multi_query', 0)
        for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
            ix.set_option(o, int(val))
        ix.search_topk(q[:512], k)
        for rep in range(2):
            for v in vals:
                ix.set_option(name, v)
                ix.set_timing(False)
                ix.search_topk(q[:256], k)
                t0 = time.perf_counter()
                ix.search_topk(q, k)
                wall = time.perf_counter() - t0
                ix.set_timing(True)
                ix.reset_stats()
                ix.search_topk(q[:1024], k)
                s = ix.stats()
                rb = ix.row_bytes
                ms = s["scan_ms"] / max(s["scan_bytes"] / float(n * rb), 1)
                print("rows %8d %s=%-3d  %.0f QPS  sweep %.1f us (%.2f TB/s)" % (
                    n, name, v, nq / wall, ms * 1e3, n * rb / ms / 1e9), flush=True)
