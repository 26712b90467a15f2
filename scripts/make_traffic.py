"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh:
per scan launch, HBM bytes = FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide streaming
reads at 64 bytes, MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both reported in KB.
    python scripts/make_traffic.py <dir with rNN_pmc_<name>_{fetch,write}_size.csv> <tag>"""
import csv, glob, json, os, re, sys
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
                if name == "cfg5radius":
                    # the 16 collect sweeps of the radius batch travel as launches of 4 + 8 + 4 (a small first and last
                    # batch): all launches of the COLLECT instantiation together are the 16 sweeps
                    if re.search(r"scan_kernel<\d+, \d+, \d+, true", r["Kernel_Name"]):
                        best = (best or 0.0) + v
                else:
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
# shared sweeps: HBM bytes per PASS from the counter summaries of scripts/pmc.sh (FETCH_SIZE x 2 + WRITE_SIZE, KB per
# launch; an int8 launch walks two passes -- two groups of 48 queries --, a bfloat16 launch one; top-k batches on
# tiled 8-bit rows take the bfloat16 sweep since round 4, mq_8bit_int8 is the int8 sweep radius batches still take)
MQ = {"mq_64bit": ("pmc_mq_bf16_64bit_sweep_summary.txt", 6144, 1), "mq_32bit": ("pmc_mq_bf16_sweep_summary.txt", 3072, 1),
      "mq_16bit": ("pmc_mq_bf16_16bit_sweep_summary.txt", 1536, 1), "mq_8bit": ("pmc_mq_bf16_8bit_sweep_summary.txt", 768, 1), "mq_8bit_int8": ("pmc_mq_i8_sweep_summary.txt", 768, 2),
      "mq_4bit": ("pmc_mq_i8_4bit_sweep_summary.txt", 384, 2)}
for key, (fname, rb, passes) in MQ.items():
    f = os.path.join(d, "%s_%s" % (tag, fname))
    if not os.path.exists(f):
        continue
    vals = {}
    for ln in open(f):
        w = ln.split()
        if len(w) >= 4 and w[0] in ("FETCH_SIZE", "WRITE_SIZE") and w[1] == "per":
            vals[w[0]] = float(w[3])
    if "FETCH_SIZE" in vals:
        hbm = (vals["FETCH_SIZE"] * 2 + vals.get("WRITE_SIZE", 0.0)) * 1024 / passes
        out[key] = {"rows": 1000000, "hbm_bytes_per_pass": int(hbm), "passes_per_launch": passes,
                    "ratio_to_algorithmic": round(hbm / (1000000.0 * rb), 4),
                    "note": "scripts/pmc.sh over scripts/dev_mq_one.py (1M x 768, 288 queries): FETCH_SIZE x 2 + WRITE_SIZE per "
                            "launch of the collect sweep, divided by its passes"}
print(json.dumps(out, indent=1))
