# kernel-time table of one dev_mqab.py run (SZG_BITS / SZG_DIM / ... from the environment) under rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
SZG_REPS=${SZG_REPS:-3} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python3 $GRAFT_REPO_ROOT/scripts/dev_mqab.py > /tmp/prof_q.log 2>&1
tail -1 /tmp/prof_q.log | cut -c1-150
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_q/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us ", r["Percentage"])
PY
