"""Summarise a rocprofv3 kernel trace: per-kernel duration and gap to the previous kernel."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", "")))
rows.sort()
by = collections.defaultdict(list)
prev_end = None
prev_by_name = {}
for s, e, n, st in rows:
    gap = (s - prev_by_name[n]) if n in prev_by_name else None
    by[n].append((e - s, gap))
    prev_by_name[n] = e
for n, v in by.items():
    d = sorted(x[0] for x in v)
    g = sorted(x[1] for x in v if x[1] is not None)
    print("%-60s n=%5d dur med %.2f us  gap-to-prev-same-kernel med %s us  p10 %s" % (
        n, len(v), d[len(d) // 2] / 1e3, ("%.2f" % (g[len(g) // 2] / 1e3)) if g else "-",
        ("%.2f" % (g[len(g) // 10] / 1e3)) if g else "-"))
