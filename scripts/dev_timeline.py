"""Print a window of a rocprofv3 kernel trace as a timeline (start, duration, kernel), times in us
relative to the first kernel of the window.  usage: dev_timeline.py <dir> [skip] [count]"""
import csv, sys, glob, re
path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"szg::\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:44], r.get("Queue_Id", "")))
rows.sort()
rows = rows[skip:skip + count]
t0 = rows[0][0]
for s, e, n, q in rows:
    print("%9.1f  +%7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
