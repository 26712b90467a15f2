"""Profiling target: a few shared sweeps.  SZG_BITS / SZG_DIM / SZG_METRIC / SZG_ROWS / SZG_NQ select the corpus."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
bits = int(os.environ.get("SZG_BITS", "32")); dim = int(os.environ.get("SZG_DIM", "768"))
metric = int(os.environ.get("SZG_METRIC", "1")); n = int(os.environ.get("SZG_ROWS", "1000000"))
nq = int(os.environ.get("SZG_NQ", "288"))
with ScanIndex(dim, bits, metric, devices=[0]) as ix:
    ix.synth(n, 1234)
    for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
        ix.set_option(o, int(val))
    q = synth_vectors(99, 0, nq, dim)
    r, d, c = ix.search_topk(q, 10)
    print("done", r[0][:3], ix.stats()["mq_launches"])
