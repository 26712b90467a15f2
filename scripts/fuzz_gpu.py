"""Differential fuzzing of the C-ABI search entry points against the CPU oracle.

Random shapes (dim, quantization, metric, rows, k, batch size), filter masks, tombstones,
two-shard handles, duplicate-heavy and scaled corpora, random tunables.  Every answer must be
the oracle's: same rows in the same order, bit-equal float64 distances (NaN == NaN).

    python scripts/fuzz_gpu.py [seconds] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from syzgydb_amd import ScanIndex

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
OPTS = [("queries_per_launch", [1, 3, 16]), ("multi_query", [0, 1]), ("force_matrix", [0, 0, 1]),
        ("serialize_scans", [0, 1]), ("mq_min", [2, 8]), ("slack", [0, 16, 40]), ("mq_hits", [64, 1024]),
        ("sketch", [0, 1, 1]), ("sketch_extra", [0, 30]), ("sketch_min_rows", [1, 1, 4096]),
        ("force_no_refine", [0, 0, 1]), ("mask_dense", [0, 1]), ("coalesce", [0, 1]), ("finish_thread", [0, 1, 1]),
        ("radius_mq", [0, 1, 1]), ("query_batch", [5, 16])]


def same(got_r, got_d, want_r, want_d):
    if len(got_r) != len(want_r):
        return False
    if [int(x) for x in got_r] != [int(x) for x in want_r]:
        return False
    g, w = np.asarray(got_d), np.asarray(want_d)
    return bool(((g == w) | (np.isnan(g) & np.isnan(w))).all())


t_end = time.time() + budget
it = fails = 0
while time.time() < t_end:
    it += 1
    bits = int(rng.choice([4, 8, 16, 32, 64]))
    metric = int(rng.integers(0, 2))
    dim = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 32, 33, 64, 100, 128, 129, 384, 500, 768, 1024]))
    n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 3000, 5000, 9000]))
    if dim * n > 4_000_000:
        n = max(1, 4_000_000 // dim)
    nq = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 33, 50]))
    if rng.random() < 0.04 and dim * n <= 400_000:   # a long call: 3+ shared-sweep batches (the finisher thread)
        nq = int(rng.choice([200, 330]))
    k = int(rng.choice([1, 2, 10, 11, 50, 100, 300, 5000]))
    kind = int(rng.integers(0, 8))
    vec = rng.uniform(-1, 1, (n, dim))
    if kind == 1:          # duplicates
        vec[rng.integers(0, n, n // 2)] = vec[0]
    elif kind == 2:        # coarse grid: many equal distances
        vec = np.round(vec * 2) / 2
    elif kind == 3 and bits >= 32:   # large magnitudes / far from the origin
        vec = vec * 1e3 + 5e3
    elif kind == 4:        # some zero rows
        vec[rng.integers(0, n, max(1, n // 10))] = 0.0
    Q = rng.uniform(-1, 1, (nq, dim))
    if kind == 5:          # rows antipodal / parallel to a query among the first rows: the unclamped acos
        for r0 in rng.integers(0, min(n, max(k, 4)), 3):   # (collection.go:831) can return NaN for them, and the
            vec[r0] = Q[0] * float(rng.choice([-0.5, -1.0, 0.5, 1.0]))   # first k rows enter the heap regardless
    elif kind == 6 and bits >= 32:   # NaN / Inf elements in stored rows
        for r0 in rng.integers(0, n, 3):
            vec[r0, int(rng.integers(0, dim))] = float(rng.choice([np.nan, np.inf, -np.inf]))
    elif kind == 7 and bits == 32:   # elements whose squares overflow float32 (finite in the reference's float64)
        for r0 in rng.integers(0, n, 4):
            vec[r0] = vec[r0] * float(rng.choice([1e19, 1e25, 1e37]))
    rows = orc.encode_rows(vec, bits)
    if kind == 3 and bits >= 32:
        Q = Q * 1e3 + 5e3
    if rng.random() < 0.2:
        Q[0] = orc.decode_vector(rows[int(rng.integers(0, n))], dim, bits)   # a stored row as the query
    if rng.random() < 0.1:
        Q[-1] = 0.0
    allow = None
    if rng.random() < 0.4:
        allow = rng.random((nq, n)) < rng.choice([0.05, 0.5, 0.95])
    devices = [0, 0] if rng.random() < 0.25 else [0]
    opts = {name: int(rng.choice(vals)) for name, vals in OPTS if rng.random() < 0.3}
    dead = []
    desc = dict(it=it, bits=bits, metric=metric, dim=dim, n=n, nq=nq, k=k, kind=kind, masked=allow is not None,
                devices=devices, opts=opts)
    try:
        with ScanIndex(dim, bits, metric, devices=devices) as ix:
            split = int(rng.integers(1, n)) if (n > 1 and rng.random() < 0.3) else n
            ix.load(rows[:split])
            if split < n:   # the rest arrives through the mutation entry points
                if rng.random() < 0.5:
                    ix.append(rows[split:])
                else:
                    ix.append_vectors(vec[split:])
            if n > 3 and rng.random() < 0.3:   # overwrite a few rows in place
                for r0 in rng.choice(n, size=min(5, n), replace=False):
                    nv = rng.uniform(-1, 1, dim)
                    vec[r0] = nv
                    rows[r0] = orc.encode_rows(nv.reshape(1, -1), bits)[0]
                    ix.overwrite(int(r0), rows[r0])
            assert (ix.read_rows(0, n) == rows).all(), "read_rows != what was written"
            for name, val in opts.items():
                ix.set_option(name, val)
            if rng.random() < 0.3 and n > 2:
                dead = [int(x) for x in rng.choice(n, size=max(1, n // 7), replace=False)]
                for r in dead:
                    ix.tombstone(r)
            live = np.ones(n, dtype=bool)
            live[dead] = False
            r, d, c = ix.search_topk(Q, k, allow=allow)
            for qi in range(nq):
                m = live if allow is None else (live & allow[qi])
                o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=k, allow=m.astype(np.uint8))
                if not same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist):
                    fails += 1
                    print("MISMATCH topk", desc, "query", qi, flush=True)
                    print("  got ", [int(x) for x in r[qi, : c[qi]]][:12], d[qi, : c[qi]][:6], flush=True)
                    print("  want", [int(x) for x in o_rows][:12], o_dist[:6], flush=True)
                    break
            # one radius search at a result distance
            qi = int(rng.integers(0, nq))
            m = live if allow is None else (live & allow[qi])
            o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], k=min(k, 20), allow=m.astype(np.uint8))
            finite = [x for x in o_dist if x == x and x > 0]
            if finite:
                radius = float(finite[-1])
                rr, dd = ix.search_radius(Q[qi], radius, allow=None if allow is None else allow[qi])
                w_rows, w_dist, _ = orc.search_exact(rows, dim, bits, metric, Q[qi], radius=radius,
                                                     allow=m.astype(np.uint8))
                if not same(rr, dd, w_rows, w_dist):
                    fails += 1
                    print("MISMATCH radius", desc, "query", qi, radius, len(rr), len(w_rows), flush=True)
            # a BATCH of radius searches, each query its own radius (query-major collect launches, szg_search_radius_batch)
            nb_ = min(nq, int(rng.choice([1, 3, 17, 40])))
            radii = []
            for qj in range(nb_):
                mj = live if allow is None else (live & allow[qj])
                od = orc.search_exact(rows, dim, bits, metric, Q[qj], k=int(rng.choice([1, 5, 60])), allow=mj.astype(np.uint8))[1]
                fin = [x for x in od if x == x and x > 0]
                radii.append(float(fin[-1]) if fin else 0.5)
            hits = ix.search_radius_batch(Q[:nb_], radii, allow=None if allow is None else allow[:nb_])
            for qj in range(nb_):
                mj = live if allow is None else (live & allow[qj])
                w_r, w_d, _ = orc.search_exact(rows, dim, bits, metric, Q[qj], radius=radii[qj], allow=mj.astype(np.uint8))
                if not same(hits[qj][0], hits[qj][1], w_r, w_d):
                    fails += 1
                    print("MISMATCH radius batch", desc, "query", qj, radii[qj], len(hits[qj][0]), len(w_r), flush=True)
                    break
            # the candidate re-rank primitives: float64 distances for row lists and row pairs
            pick = rng.integers(0, n, min(n, 40)).astype(np.uint64)
            got = ix.distances(Q[0], pick)
            want = orc.all_distances(rows[pick.astype(np.int64)], dim, bits, metric, Q[0])
            if not bool(((got == want) | (np.isnan(got) & np.isnan(want))).all()):
                fails += 1
                print("MISMATCH distances", desc, flush=True)
            pa = rng.integers(0, n, 20).astype(np.uint64)
            pb = rng.integers(0, n, 20).astype(np.uint64)
            got = ix.pair_distances(pa, pb)
            fn = orc.angular if metric == 1 else orc.euclidean
            want = np.array([fn(orc.decode_vector(rows[int(x)], dim, bits), orc.decode_vector(rows[int(y)], dim, bits))
                             for x, y in zip(pa, pb)])
            if not bool(((got == want) | (np.isnan(got) & np.isnan(want))).all()):
                fails += 1
                print("MISMATCH pair_distances", desc, flush=True)
            # a radius search into a buffer that is too small: the best `cap` hits and the true total
            if finite and len(w_rows) > 3:
                cap = int(rng.integers(1, len(w_rows)))
                tr, td, total = ix.search_radius(Q[qi], radius, allow=None if allow is None else allow[qi], capacity=cap)
                if total != len(w_rows) or not same(tr, td, w_rows[:cap], w_dist[:cap]):
                    fails += 1
                    print("MISMATCH truncated radius", desc, cap, total, len(w_rows), flush=True)
            # a second act: mutations AFTER searches (whatever the handle keeps beside the rows -- resident row norms,
            # sketches, liveness words -- has to follow them), then the same questions again
            if n > 3 and allow is None and rng.random() < 0.4:
                for r0 in rng.choice(n, size=min(4, n), replace=False):
                    if r0 in dead:
                        continue
                    nv = rng.uniform(-1, 1, dim) * float(rng.choice([1.0, 0.3, 0.0]))
                    if rng.random() < 0.5:
                        rows[r0] = orc.encode_rows(nv.reshape(1, -1), bits)[0]
                        ix.overwrite(int(r0), rows[r0])
                    else:
                        ix.overwrite_vector(int(r0), nv)
                        rows[r0] = ix.read_rows(int(r0), 1)[0]
                        assert (rows[r0] == orc.encode_rows(nv.reshape(1, -1), bits)[0]).all(), "device quantization differs"
                extra = int(rng.integers(1, 40))
                ev = rng.uniform(-1, 1, (extra, dim))
                erows = orc.encode_rows(ev, bits)
                ix.append(erows)
                rows2 = np.concatenate([rows, erows])
                live2 = np.concatenate([live, np.ones(extra, dtype=bool)])
                r, d, c = ix.search_topk(Q, k)
                for qi in range(nq):
                    o_rows, o_dist, _ = orc.search_exact(rows2, dim, bits, metric, Q[qi], k=k, allow=live2.astype(np.uint8))
                    if not same(r[qi, : c[qi]], d[qi, : c[qi]], o_rows, o_dist):
                        fails += 1
                        print("MISMATCH topk after mutation", desc, "query", qi, flush=True)
                        break
                nb_ = min(nq, 9)
                radii = []
                for qj in range(nb_):
                    od = orc.search_exact(rows2, dim, bits, metric, Q[qj], k=5, allow=live2.astype(np.uint8))[1]
                    fin = [x for x in od if x == x and x > 0]
                    radii.append(float(fin[-1]) if fin else 0.5)
                hits = ix.search_radius_batch(Q[:nb_], radii)
                for qj in range(nb_):
                    w_r, w_d, _ = orc.search_exact(rows2, dim, bits, metric, Q[qj], radius=radii[qj], allow=live2.astype(np.uint8))
                    if not same(hits[qj][0], hits[qj][1], w_r, w_d):
                        fails += 1
                        print("MISMATCH radius batch after mutation", desc, "query", qj, flush=True)
                        break
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("EXCEPTION", desc, repr(e), flush=True)
    if it % 25 == 0:
        print("iterations %d, failures %d" % (it, fails), flush=True)
print("FUZZ done: %d iterations, %d failures (seed %d)" % (it, fails, seed), flush=True)
sys.exit(1 if fails else 0)
