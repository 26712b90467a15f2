"""Profiling target: a few 32-query shared sweeps over 1M x 768 f32 cosine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
with ScanIndex(768, 32, 1, devices=[0]) as ix:
    ix.synth(1000000, 1234)
    q = synth_vectors(99, 0, 256, 768)
    r, d, c = ix.search_topk(q, 10)
    print("done", r[0][:3], ix.stats()["mq_launches"])
