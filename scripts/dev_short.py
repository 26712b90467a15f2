"""Short calls on a small shard (the driver's N=8 form: 20 queries per timed call on 125 K rows): wall time per
call, sweeps, and what is NOT a sweep (pipeline fill / drain, host work).  SZG_OPTS=name=v,... applies tunables;
SZG_CALL = queries per call (20), SZG_ROWS (125056)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
dim, bits, metric, k = 768, 32, 1, int(os.environ.get("SZG_K", "10"))
n = int(os.environ.get("SZG_ROWS", "125056"))
per = int(os.environ.get("SZG_CALL", "20"))
q = synth_vectors(99, 0, 64 * per, dim)
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    ix.set_option('multi_query', 0)
    for name, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(name, int(val))
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        ix.search_topk(q[:per], k)
    ix.set_timing(True)
    walls, sweeps = [], []
    for i in range(64):
        ix.reset_stats()
        t0 = time.perf_counter()
        ix.search_topk(q[i * per:(i + 1) * per], k)
        walls.append(1e3 * (time.perf_counter() - t0))
        sweeps.append(ix.stats()["scan_ms"])
    walls.sort(); sweeps.sort()
    w, s = walls[len(walls) // 2], sweeps[len(sweeps) // 2]
    print("%s rows %d, %d queries per call: median wall %.3f ms = %.0f q/s, sweeps %.3f ms, fixed overhead %.3f ms" % (
        os.environ.get("SZG_OPTS", "-"), n, per, w, per / w * 1e3, s, w - s), flush=True)
