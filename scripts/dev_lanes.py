"""Dev probe: scan GB/s vs lane-group width L for a config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits, metric = [int(x) for x in sys.argv[1:5]]
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    q = synth_vectors(99, 0, 80, dim)
    ix.set_option('multi_query', 0)
    ix.set_timing(True)
    for L in [0] + [int(x) for x in sys.argv[5:]]:
        if L: ix.set_option('lanes_per_row', L)
        ix.search_topk(q[:16], 10)
        ix.reset_stats()
        ix.search_topk(q[16:], 10)
        st = ix.stats()
        ms = st['scan_ms'] / st['timed_launches']
        print("%dx%d q%d m%d L=%s: scan %.3f ms = %.2f TB/s" % (n, dim, bits, metric, L or 'auto', ms, ix.rows * ix.row_bytes / ms / 1e9), flush=True)
