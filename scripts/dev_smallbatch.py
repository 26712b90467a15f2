"""Latency of small batches: one sweep per query vs the shared sweep (mq_min)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
n, dim, metric, k = 1000000, 768, 1, 10
for bits in (32, 8, 4):
    q = synth_vectors(99, 0, 64, dim)
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.synth(n, 1234)
        for mq_min in (8, 2):
            ix.set_option("mq_min", mq_min)
            line = "bits=%d mq_min=%d:" % (bits, mq_min)
            for nq in (1, 2, 3, 4, 6, 8):
                ix.search_topk(q[:nq], k)
                t0 = time.perf_counter()
                for rep in range(20):
                    ix.search_topk(q[rep:rep + nq], k)
                line += "  nq=%d %.0f us" % (nq, (time.perf_counter() - t0) / 20 * 1e6)
            print(line, flush=True)
