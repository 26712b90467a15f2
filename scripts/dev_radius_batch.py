"""cfg5-shard radius batches: wall per call against the sweeps' own time.  SZG_NQ (24), SZG_ROWS (12500000)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim, bits, metric = int(os.environ.get("SZG_DIM", "384")), int(os.environ.get("SZG_BITS", "4")), 1
n = int(os.environ.get("SZG_ROWS", "12500000"))
nq = int(os.environ.get("SZG_NQ", "24"))
q = synth_vectors(7, 0, max(nq, 16), dim)
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    for name, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(name, int(val))
    _, d, _ = ix.search_topk(q[0], 500)
    radius = float(d[0, -1])
    for _ in range(4):
        ix.search_radius_batch(q[:nq], radius)
    ix.set_timing(True)
    for rep in range(4):
        ix.reset_stats()
        t0 = time.perf_counter()
        hits = ix.search_radius_batch(q[:nq], radius)
        wall = 1e3 * (time.perf_counter() - t0)
        s = ix.stats()
        print("nq %d: wall %.3f ms (%.0f q/s, %.2f TB/s), sweeps %.3f ms in %d launches (%.2f TB/s), rest %.3f ms; host us: prep %.0f enq %.0f fin %.0f; hits/query %.0f" % (
            nq, wall, nq / wall * 1e3, nq * n * ix.row_bytes / wall / 1e9, s["scan_ms"], s["timed_launches"],
            s["scan_bytes"] / s["scan_ms"] / 1e9, wall - s["scan_ms"], s["host_prep_us"], s["host_enqueue_us"], s["host_finish_us"],
            sum(len(r) for r, _ in hits) / nq), flush=True)
