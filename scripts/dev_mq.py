"""Dev probe: multi-query sweep time vs batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim = 1000000, 768
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 3: n, dim = int(sys.argv[2]), int(sys.argv[3])
with ScanIndex(dim, bits, 1, devices=[0]) as ix:
    ix.synth(n, 1234)
    q = synth_vectors(99, 0, 1024, dim)
    ix.set_timing(True)
    ix.search_topk(q[:480], 10)
    for B in (48,):
        # batches of exactly B: feed B queries per call
        ix.search_topk(q[:B], 10)
        ix.reset_stats()
        t0 = time.time()
        for i in range(0, 16 * B, B):
            ix.search_topk(q[i:i + B], 10)
        dt = time.time() - t0
        st = ix.stats()
        ms = st['scan_ms'] / st['timed_launches']
        print("B=%2d: sweep %.3f ms (%.2f TB/s alg, %.1f TFLOP/s) sync-call QPS %.0f pipeline %.3f ms/batch" % (
            B, ms, ix.rows * ix.row_bytes / ms / 1e9, 2.0 * n * dim * B / ms / 1e9, 16 * B / dt, st['total_ms'] / 16), flush=True)
    ix.search_topk(q[:1008], 10)
    ix.reset_stats()
    t0 = time.time(); ix.search_topk(q[:1008], 10); dt = time.time() - t0
    st = ix.stats()
    print("bits=%d %dx%d: 1008 queries pipelined: %.0f QPS, sweep %.3f ms" % (bits, n, dim, 1008 / dt, st['scan_ms'] / st['timed_launches']))
