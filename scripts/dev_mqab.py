"""Shared-sweep throughput of the loaded library (A/B via SZG_LIB_PATH)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, bits, metric, k = int(os.environ.get('SZG_ROWS', '1000000')), int(os.environ.get('SZG_DIM', '768')), int(os.environ.get('SZG_BITS', '32')), int(os.environ.get('SZG_METRIC', '1')), 10
nq = int(os.environ.get('SZG_NQ', '960'))
q = synth_vectors(99, 0, nq, dim)
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(o, int(val))
    ix.search_topk(q, k); ix.search_topk(q, k)
    sweeps, walls = [], []
    for rep in range(int(os.environ.get('SZG_REPS', '8'))):
        ix.set_timing(False)
        t0 = time.perf_counter(); ix.search_topk(q, k); walls.append(time.perf_counter() - t0)
        ix.set_timing(True); ix.reset_stats(); ix.search_topk(q, k); s = ix.stats()
        sweeps.append(s["scan_ms"] / max(s["timed_launches"], 1))
    sweeps.sort(); walls.sort()
    ms, wall = sweeps[len(sweeps) // 2], walls[len(walls) // 2]
    per = s["mq_queries"] / max(s["mq_launches"], 1)
    print("%s %s bits=%d dim=%d metric=%d: %.0f QPS  sweep min %.4f med %.4f ms  %.2f TB/s  (%.1f q/pass)  esc %d fb %d  host us/query: prep %.2f enq %.2f fin %.2f" % (
        os.path.basename(os.environ.get("SZG_LIB_PATH", "default")), os.environ.get("SZG_OPTS", ""), bits, dim, metric, nq / wall, sweeps[0], ms,
        n * dim * bits / 8 / (ms * 1e-3) / 1e12, per, s["escalations"], s["mq_fallbacks"],
        s["host_prep_us"] / nq, s["host_enqueue_us"] / nq, s["host_finish_us"] / nq), flush=True)
