#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline object into
# gpurun_out/profiles/ (copy what is to be judged into profiles/).  Run on the GPU box
# from the repo root:  bash scripts/collect_profiles.sh r01
set -e
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
one() { f=$(find "$1" -name "$2" | head -1); if [ -n "$f" ]; then cp "$f" "$3"; else echo "missing $2 under $1" >&2; fi; }

echo "[1/5] unprofiled bench" | tee -a $out/progress.log
python3 $GRAFT_REPO_ROOT/bench.py > $out/${tag}_bench_n1_unprofiled_stdout.json 2> $out/bench_unprofiled.err

echo "[2/5] kernel trace + stats over bench.py" | tee -a $out/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py > $out/${tag}_bench_n1_stdout.json 2> $out/bench_profiled.err
one /tmp/prof_stats "*kernel_stats.csv" $out/${tag}_bench_n1_kernel_stats.csv

i=2
for cfg in "headline 1000000 768 32 1 10 16" "cfg3 1000000 768 8 1 10 16"; do
  set -- $cfg; name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    echo "[$i] pmc $ctr $name" | tee -a $out/progress.log
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_${name}_$ctr -- python3 $GRAFT_REPO_ROOT/scripts/dev_one.py "$@" > /tmp/pmc.log 2>&1
    lc=$(echo $ctr | tr A-Z a-z)
    one /tmp/prof_${name}_$ctr "*counter_collection.csv" $out/${tag}_pmc_${name}_$lc.csv
  done
done
echo done | tee -a $out/progress.log
