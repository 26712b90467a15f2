"""One removed document must not slow the sweep down (masked sweeps with the dense phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, k = 1000000, 768, 10
q = synth_vectors(99, 0, 1024, dim)
with ScanIndex(dim, 32, 1, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.set_option("multi_query", 0)
    for label in ("no tombstones", "one tombstone", "one tombstone, mask_dense=0"):
        if label == "one tombstone":
            ix.tombstone(12345)
        if label.endswith("=0"):
            ix.set_option("mask_dense", 0)
        ix.search_topk(q[:256], k)
        ix.set_timing(True); ix.reset_stats()
        t0 = time.perf_counter(); ix.search_topk(q, k); el = time.perf_counter() - t0
        s = ix.stats(); ix.set_timing(False)
        ms = s["scan_ms"] / max(s["scan_bytes"] / (n * 3072.0), 1)
        print("%-30s %.0f queries/s, sweep %.1f us (%.2f TB/s)" % (label, 1024 / el, ms * 1e3, n * 3072 / ms / 1e9), flush=True)
