"""Stateful fuzzing of the Python mirror of the reference's Collection API (add / re-add /
remove / update / search / list) against a dict model scored by the oracle.

    python scripts/fuzz_collection.py [seconds] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from syzgydb_amd import Collection, CollectionOptions, SearchArgs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    cases += 1
    dim = int(rng.choice([1, 3, 16, 64, 100, 128]))
    bits = int(rng.choice([4, 8, 16, 32, 64]))
    metric = int(rng.integers(0, 2))
    devices = [0, 0] if rng.random() < 0.3 else [0]
    c = Collection(CollectionOptions(Name="fuzz", DistanceMethod=metric, DimensionCount=dim, Quantization=bits),
                   devices=devices)
    order = []     # row order: ids as first added (None once removed)
    model = {}     # id -> (packed row bytes, metadata)
    try:
        for step in range(int(rng.integers(5, 120))):
            op = rng.random()
            if op < 0.5 or not model:
                id = int(rng.choice([int(rng.integers(0, 40)), int(rng.integers(0, 10**12))]))
                v = rng.uniform(-1.2, 1.2, dim)
                meta = bytes(rng.integers(0, 256, int(rng.integers(0, 12))).astype(np.uint8))
                if rng.random() < 0.3 and id not in model:
                    c.AddDocuments([id], v.reshape(1, -1), [meta])
                else:
                    c.AddDocument(id, v, meta)
                if id not in model:
                    order.append(id)
                model[id] = (orc.encode_rows(v.reshape(1, -1), bits)[0], meta)
            elif op < 0.6:
                id = int(rng.choice(list(model)))
                c.removeDocument(id)
                order[order.index(id)] = None
                del model[id]
            elif op < 0.7:
                id = int(rng.choice(list(model)))
                meta = b"upd%d" % step
                c.UpdateDocument(id, meta)
                model[id] = (model[id][0], meta)
            else:
                live = sorted(model, key=str)   # the reference's deterministic visit order (spanfile.go:540-560)
                rows = np.stack([model[i][0] for i in live]) if live else np.zeros((0, 1), np.uint8)
                flt = None
                if rng.random() < 0.4:
                    mod = int(rng.integers(2, 4))
                    flt = lambda id, md, mod=mod: id % mod == 0 or md.startswith(b"upd")   # noqa: E731
                allow = np.array([1 if (flt is None or flt(i, model[i][1])) else 0 for i in live], np.uint8)
                q = rng.uniform(-1, 1, dim)
                mode = rng.random()
                if mode < 0.15:     # listing mode (collection.go:633-669): string-sorted ids, Offset / Limit
                    off, lim = int(rng.integers(0, 3)), int(rng.integers(0, 4))
                    res = c.Search(SearchArgs(Filter=flt, Offset=off, Limit=lim))
                    ids = sorted([i for i, a in zip(live, allow) if a], key=str)[off:]
                    if lim > 0:
                        ids = ids[:lim]
                    if [r.ID for r in res.Results] != ids:
                        fails += 1
                        print("MISMATCH listing", cases, step, [r.ID for r in res.Results][:6], ids[:6], flush=True)
                        break
                else:
                    k = int(rng.choice([1, 3, 10, 200]))
                    radius = 0.0
                    if mode < 0.4 and live:
                        radius = float(rng.choice([0.2, 0.5, 1.5, 5.0]))
                    res = c.Search(SearchArgs(Vector=q, Filter=flt, K=k, Radius=radius, Precision="exact"))
                    if live:
                        o_rows, o_dist, _ = orc.search_exact(rows, dim, bits, metric, q, k=k, radius=radius, allow=allow)
                    else:
                        o_rows, o_dist = [], []
                    got = [(r.ID, r.Distance, r.Metadata) for r in res.Results]
                    want = [(live[int(r)], float(d), model[live[int(r)]][1]) for r, d in zip(o_rows, o_dist)]
                    ok = len(got) == len(want) and all(
                        g[0] == w[0] and (g[1] == w[1] or (g[1] != g[1] and w[1] != w[1])) and g[2] == w[2]
                        for g, w in zip(got, want))
                    ok = ok and res.PercentSearched == (100.0 if live else 0.0)
                    if not ok:
                        fails += 1
                        print("MISMATCH search", dict(case=cases, step=step, dim=dim, bits=bits, metric=metric, k=k,
                                                      radius=radius, n=len(live), devices=devices), flush=True)
                        print("  got ", got[:5], "\n  want", want[:5], flush=True)
                        break
            if rng.random() < 0.1 and model:
                id = int(rng.choice(list(model)))
                d = c.GetDocument(id)
                want_v = orc.decode_vector(model[id][0], dim, bits)
                if not (np.array_equal(np.asarray(d.Vector), want_v) and d.Metadata == model[id][1]):
                    fails += 1
                    print("MISMATCH GetDocument", cases, step, flush=True)
                    break
        assert c.GetDocumentCount() == len(model) and c.GetAllIDs() == sorted(model)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("EXCEPTION", dict(case=cases, dim=dim, bits=bits, metric=metric, devices=devices), repr(e), flush=True)
    finally:
        c.Close()
print("COLLECTION FUZZ done: %d cases, %d failures (seed %d)" % (cases, fails, seed), flush=True)
sys.exit(1 if fails else 0)
