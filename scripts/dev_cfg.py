"""Kernel-only sweep rate of the single-query scan for a list of configs, for the library
selected by SZG_LIB_PATH (A/B of builds: run once per library inside ONE gpurun call).
  python scripts/dev_cfg.py rows:dim:bits:metric:k[:radius] ...      SZG_OPTS=name=v,... applies options"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syzgydb_amd import ScanIndex
from syzgydb_amd.synth import synth_vectors
tag = os.path.basename(os.environ.get("SZG_LIB_PATH", "default")).replace("libsyzgy_scan_", "").replace(".so", "")
nq = int(os.environ.get("SZG_NQ", "256"))
for spec in sys.argv[1:]:
    f = spec.split(":")
    n, dim, bits, metric, k = [int(x) for x in f[:5]]
    radius = float(f[5]) if len(f) > 5 else 0.0
    q = synth_vectors(99, 0, nq, dim)
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.synth(n, 1234)
        ix.set_option('multi_query', 0)
        for o, val in [x.split('=') for x in os.environ.get('SZG_OPTS', '').split(',') if x]:
            ix.set_option(o, int(val))
        if radius > 0:
            _, dd, _ = ix.search_topk(q[0], 500)
            radius = float(dd[0, -1])
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=3)   # concurrent callers keep the sweeps back to back
            run = lambda qq: list(pool.map(lambda x: ix.search_radius(x, radius), qq[:96]))
        else:
            run = lambda qq: ix.search_topk(qq, k)
        run(q[:64]); run(q)
        best = None
        for rep in range(3):
            ix.set_timing(True); ix.reset_stats(); run(q); s = ix.stats(); ix.set_timing(False)
            sweeps = s["scan_bytes"] / float(n * ix.row_bytes)
            us = 1e3 * s["scan_ms"] / max(sweeps, 1)
            best = us if best is None else min(best, us)
        print("%-10s %9d x %4d %2d-bit m%d %s: %8.1f us/sweep  %.2f TB/s" % (
            tag, n, dim, bits, metric, "radius" if radius > 0 else "k=%d" % k, best, n * ix.row_bytes / best / 1e6), flush=True)
