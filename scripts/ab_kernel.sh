# Average duration (rocprofv3 --kernel-trace --stats) of the kernels matching $PAT for every library
# variant, on one box.   PAT=mq_score_i8 VARIANTS="default a b" [SZG_BITS=8 ...] bash scripts/ab_kernel.sh script.py [args]
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset SZG_LIB_PATH; else export SZG_LIB_PATH=$GRAFT_REPO_ROOT/syzgydb_amd/variants/libsyzgy_scan_$v.so; fi
  rm -rf /tmp/abk
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $GRAFT_REPO_ROOT/scripts/$1 ${@:2} > /tmp/abk.log 2>&1 || { echo "$v: run failed"; tail -3 /tmp/abk.log; continue; }
  f=$(find /tmp/abk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$PAT" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        print("%-8s %-90s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (sys.argv[3], r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
done
unset SZG_LIB_PATH
