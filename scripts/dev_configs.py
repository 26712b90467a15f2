"""Dev probe: scan-kernel GB/s for each BASELINE config (top-k form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors

CFG = [("headline", 1_000_000, 768, 32, 1, 10), ("cfg2", 1_000_000, 384, 32, 1, 10),
       ("cfg3 q8", 1_000_000, 768, 8, 1, 10), ("cfg4 shard", 1_250_000, 768, 32, 0, 100),
       ("cfg5 q4 slice", 12_500_000, 384, 4, 1, 10), ("q16", 1_000_000, 768, 16, 1, 10),
       ("q64", 500_000, 768, 64, 1, 10), ("q8 euclid", 1_000_000, 768, 8, 0, 10),
       ("q4 euclid", 4_000_000, 768, 4, 0, 10)]
only = sys.argv[1:] 
for name, n, dim, bits, metric, k in CFG:
    if only and not any(o in name for o in only): continue
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.synth(n, 1234)
        q = synth_vectors(99, 0, 96, dim)
        ix.search_topk(q[:16], k)
        ix.set_timing(True); ix.reset_stats()
        t0 = time.time(); ix.search_topk(q[16:], k); dt = time.time() - t0
        st = ix.stats()
        ms = st['scan_ms'] / st['timed_launches']
        gb = ix.rows * ix.row_bytes
        print("%-14s %9d x %4d q%-2d m%d k%-3d: %8.1f QPS | scan %.3f ms = %.2f TB/s (%.0f%% of 8) esc=%d"
              % (name, n, dim, bits, metric, k, 80 / dt, ms, gb / ms / 1e9, gb / ms / 1e9 / 80, st['escalations']), flush=True)
