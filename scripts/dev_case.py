"""A hand-built search case against the oracle (for chasing a fuzz mismatch): far-from-origin 32-bit rows of 2 dims."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from syzgydb_amd import ScanIndex
dim, bits, metric, n, nq, k = int(os.environ.get("DIM", 2)), 32, 1, 5000, 8, int(os.environ.get("K", 1))
bad = 0
for seed in range(int(os.environ.get("SEEDS", 12))):
    rng = np.random.default_rng(seed)
    vec = rng.uniform(-1, 1, (n, dim)) * 1e3 + 5e3
    Q = rng.uniform(-1, 1, (nq, dim)) * 1e3 + 5e3
    rows = orc.encode_rows(vec, bits)
    allow = rng.random((nq, n)) < float(os.environ.get("PASS", 0.5))
    with ScanIndex(dim, bits, metric, devices=[0]) as ix:
        ix.load(rows)
        for o in [x for x in os.environ.get("SZG_OPTS", "mq_fused=0,mq_min=8").split(",") if x]:
            name, val = o.split("=")
            ix.set_option(name, int(val))
        r, d, c = ix.search_topk(Q, k, allow=allow)
        st = ix.stats()
        for qi in range(nq):
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=k, allow=allow[qi].astype(np.uint8))
            ok = list(map(int, r[qi, :c[qi]])) == list(map(int, o_rows)) and bool((d[qi, :c[qi]] == np.asarray(o_dist)).all())
            if not ok:
                bad += 1
                print("seed", seed, "query", qi, "got", r[qi, :c[qi]], d[qi, :c[qi]], "want", o_rows, o_dist,
                      {x: st[x] for x in ("escalations", "mq_fallbacks", "full_replays", "mq_bf16_sweeps", "mq_launches")})
print("mismatches:", bad)
