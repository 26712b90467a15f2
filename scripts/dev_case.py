"""A hand-built search case against the oracle (for chasing a fuzz mismatch): radius batches on 16-bit rows of few dims."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from syzgydb_amd import ScanIndex
dim, bits, metric, n, nq = int(os.environ.get("DIM", 1)), int(os.environ.get("BITS", 16)), int(os.environ.get("METRIC", 0)), 3000, 9
bad = 0
for seed in range(int(os.environ.get("SEEDS", 8))):
    rng = np.random.default_rng(seed)
    vec = rng.uniform(-1, 1, (n, dim))
    Q = rng.uniform(-1, 1, (nq, dim))
    rows = orc.encode_rows(vec, bits)
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.load(rows)
        for o in [x for x in os.environ.get("SZG_OPTS", "").split(",") if x]:
            name, val = o.split("=")
            ix.set_option(name, int(val))
        radii = []
        for qj in range(nq):
            od = orc.search_exact(rows, dim, bits, metric, Q[qj], k=[1, 5, 60][qj % 3])[1]
            fin = [x for x in od if x == x and x > 0]
            radii.append(float(fin[-1]) if fin else 0.5)
        hits = ix.search_radius_batch(Q, radii)
        st = ix.stats()
        for qj in range(nq):
            w_r, w_d, _ = orc.search_exact(rows, dim, bits, metric, Q[qj], radius=radii[qj])
            ok = list(map(int, hits[qj][0])) == list(map(int, w_r)) and bool((np.asarray(hits[qj][1]) == np.asarray(w_d)).all())
            if not ok:
                bad += 1
                print("seed", seed, "query", qj, "radius", radii[qj], "got", len(hits[qj][0]), "want", len(w_r),
                      {x: st[x] for x in ("mq_queries", "mq_bf16_sweeps", "escalations")})
print("mismatches:", bad)
