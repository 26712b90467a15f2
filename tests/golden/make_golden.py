#!/usr/bin/env python3
"""Generates tests/golden/scan_cases.json from the CPU oracle.

The reference is Go and cannot run in this image, so these vectors are outputs
of oracle/ (a line-by-line C restatement of the reference's scan path) on
seeded synthetic inputs -- they pin the HIP path and any later oracle change
against today's oracle.  The reference's OWN known answers live in
reference_kats.json (transcribed from its tests, see that file).

Floats are stored as IEEE-754 hex bit patterns so the comparison is bit-exact.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402

SEED = 0x53595A4700000000


def f64hex(a):
    return [struct.pack(">d", float(x)).hex() for x in np.asarray(a, dtype=np.float64).ravel()]


def main():
    cases = []
    cid = 0
    for bits in (4, 8, 16, 32, 64):
        for metric in (0, 1):
            for dim, n in ((3, 40), (16, 120), (100, 65)):
                cid += 1
                seed = SEED + cid
                rows = orc.synth_rows(seed, 0, n, dim, bits)
                q = orc.synth_vectors(seed + 1000, 0, 1, dim)[0]
                if dim == 3:  # the reference's own tests use positive vectors at dim 3
                    q = np.abs(q)
                alld = orc.all_distances(rows, dim, bits, metric, q)
                radius = float(np.sort(alld)[min(n - 1, 12)])
                allow = (np.arange(n) % 3 != 0).astype(np.uint8)
                t_rows, t_dist, _ = orc.search_exact(rows, dim, bits, metric, q, k=7)
                r_rows, r_dist, _ = orc.search_exact(rows, dim, bits, metric, q, radius=radius)
                f_rows, f_dist, _ = orc.search_exact(rows, dim, bits, metric, q, k=5, allow=allow)
                cases.append({
                    "id": cid, "bits": bits, "metric": metric, "dim": dim, "n": n, "seed": seed,
                    # small corpora are stored; larger ones are regenerated from the seed
                    # (orc_synth_rows) and pinned by their SHA-256
                    "rows_hex": rows.tobytes().hex() if rows.size <= 4096 else None,
                    "rows_sha256": hashlib.sha256(rows.tobytes()).hexdigest(),
                    "query_hex": f64hex(q),
                    "topk": {"k": 7, "rows": [int(x) for x in t_rows], "dist_hex": f64hex(t_dist)},
                    "radius": {"radius_hex": f64hex([radius])[0], "rows": [int(x) for x in r_rows],
                               "dist_hex": f64hex(r_dist)},
                    "filtered": {"k": 5, "allow_mod3_ne0": True, "rows": [int(x) for x in f_rows],
                                 "dist_hex": f64hex(f_dist)},
                })
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scan_cases.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "cases": cases}, f, indent=0)
    print("wrote", out, os.path.getsize(out), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
