"""Dev probe: solo scan-kernel time / GB/s for one configuration and geometry sweep."""
import argparse, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=1000000)
ap.add_argument('--dim', type=int, default=768)
ap.add_argument('--bits', type=int, default=32)
ap.add_argument('--metric', type=int, default=1)
ap.add_argument('--k', type=int, default=10)
ap.add_argument('--queries', type=int, default=48)
ap.add_argument('--geoms', default='1x4x256')
ap.add_argument('--mq', type=int, default=0)
a = ap.parse_args()
with ScanIndex(a.dim, a.bits, a.metric, devices=[0]) as ix:
    ix.synth(a.n, 1234)
    q = synth_vectors(99, 0, a.queries, a.dim)
    ix.set_timing(True)
    ix.set_option('multi_query', a.mq)
    for g in a.geoms.split(','):
        c, bpc, bt = map(int, g.split('x'))
        ix.set_option('contexts', c); ix.set_option('blocks_per_cu', bpc); ix.set_option('block_threads', bt)
        ix.search_topk(q[:4], a.k)
        ix.reset_stats()
        t0 = time.time(); ix.search_topk(q, a.k); dt = time.time() - t0
        st = ix.stats()
        ms = st['scan_ms'] / st['timed_launches']
        print("ctx=%d blocks/cu=%d threads=%d mq=%d: wall %.1f QPS | sweep %.3f ms/launch = %.2f TB/s | launches %d mq_queries %d esc=%d"
              % (c, bpc, bt, a.mq, a.queries / dt, ms,
                 ix.rows * ix.row_bytes / ms / 1e9, st['scan_launches'], st['mq_queries'], st['escalations']))
