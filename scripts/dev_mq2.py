import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
with ScanIndex(768, 32, 1, devices=[0]) as ix:
    ix.synth(1000000, 1234)
    q = synth_vectors(99, 0, 1088, 768)
    ix.set_option('multi_query', 0)
    ix.search_topk(q[:64], 10)
    t0 = time.time(); ix.search_topk(q[64:], 10); print("single 1024: %.0f QPS" % (1024 / (time.time() - t0)))
    ix.set_option('multi_query', 1)
    ix.search_topk(q[:192], 10); ix.search_topk(q[:16], 10)
    for rep in range(4):
        ix.set_timing(rep % 2 == 0)
        t0 = time.time(); ix.search_topk(q[64:], 10); dt = time.time() - t0
        print("mq 1024 rep %d timing=%d: %.0f QPS (%.1f ms)" % (rep, rep % 2 == 0, 1024 / dt, dt * 1e3))
    t0 = time.time(); ix.search_topk(q[:1056], 10); dt = time.time() - t0
    print("mq 1056: %.0f QPS" % (1056 / dt))
