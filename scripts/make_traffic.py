"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh:
per scan launch, HBM bytes = FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide streaming
reads at 64 bytes, MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both reported in KB.
    python scripts/make_traffic.py <dir with rNN_pmc_<name>_{fetch,write}_size.csv> <tag>"""
import csv, glob, json, os, sys
d, tag = sys.argv[1], sys.argv[2]
ALG = {"headline": (1000000, 3072), "cfg2": (1000000, 1536), "cfg3": (1000000, 768), "cfg4shard": (1250000, 3072),
       "cfg5shard": (12500000, 192), "cfg5radius": (12500000, 192)}
alias = {"cfg4shard": "cfg4", "cfg5shard": "cfg5", "cfg5radius": "cfg5_radius"}
out = {}
for name, (rows, rb) in ALG.items():
    vals = {}
    for ctr in ("fetch_size", "write_size"):
        f = os.path.join(d, "%s_pmc_%s_%s.csv" % (tag, name, ctr))
        if not os.path.exists(f):
            continue
        best = None  # the launch that walks all 16 sweeps = the largest value among the scan launches
        for r in csv.DictReader(open(f)):
            if "scan_kernel" in r["Kernel_Name"]:
                v = float(r["Counter_Value"])
                best = v if best is None else max(best, v)
        if best is not None:
            vals[ctr] = best
    if "fetch_size" in vals:
        hbm = vals["fetch_size"] * 2 * 1024 + vals.get("write_size", 0.0) * 1024
        alg = 16.0 * rows * rb
        out[alias.get(name, name)] = {
            "fetch_size_kb_raw": vals["fetch_size"], "write_size_kb": vals.get("write_size"),
            "hbm_bytes_per_launch": int(hbm), "sweeps_per_launch": 16, "rows": rows,
            "ratio_to_algorithmic": round(hbm / alg, 4),
            "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over "
                    "scripts/dev_one.py (ONE query-major scan launch of 16 sweeps); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 counts the 128-byte requests of wide streaming reads at 64 bytes)"}
print(json.dumps(out, indent=1))
