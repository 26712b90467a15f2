#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline objects into gpurun_out/profiles/
# (copy what is to be judged into profiles/).  Run on the GPU box from the repo root:
#     bash scripts/collect_profiles.sh r02
# Counters are collected in their own runs (--kernel-trace --pmc only, one counter set per run).
set -e
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
one() { f=$(find "$1" -name "$2" | head -1); if [ -n "$f" ]; then cp "$f" "$3"; else echo "missing $2 under $1" >&2; fi; }

echo "[1] unprofiled bench" | tee -a $out/progress.log
python3 $GRAFT_REPO_ROOT/bench.py > $out/${tag}_bench_n1_unprofiled_stdout.json 2> $out/bench_unprofiled.err
grep "^bench detail: " $out/bench_unprofiled.err | sed "s/^bench detail: //" > $out/${tag}_bench_n1_unprofiled_detail.json || true

echo "[2] kernel trace + stats over bench.py" | tee -a $out/progress.log
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py > $out/${tag}_bench_n1_stdout.json 2> $out/bench_profiled.err
one /tmp/prof_stats "*kernel_stats.csv" $out/${tag}_bench_n1_kernel_stats.csv
grep "^bench detail: " $out/bench_profiled.err | sed "s/^bench detail: //" > $out/${tag}_bench_n1_detail.json || true

i=2
# name rows dim bits metric k queries (one query-major launch of 16 sweeps)
for cfg in "headline 1000000 768 32 1 10 16" "cfg2 1000000 384 32 1 10 16" "cfg3 1000000 768 8 1 10 16" \
           "cfg4shard 1250000 768 32 0 100 16" "cfg5shard 12500000 384 4 1 10 16"; do
  set -- $cfg; name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    echo "[$i] pmc $ctr $name" | tee -a $out/progress.log
    rm -rf /tmp/prof_${name}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_${name}_$ctr -- python3 $GRAFT_REPO_ROOT/scripts/dev_one.py "$@" > /tmp/pmc.log 2>&1
    lc=$(echo $ctr | tr A-Z a-z)
    one /tmp/prof_${name}_$ctr "*counter_collection.csv" $out/${tag}_pmc_${name}_$lc.csv
  done
done

# the radius (collect) form of cfg5's shard: one query-major launch of 16 collect sweeps, ~500 hits per query
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "[pmc] $ctr cfg5shard radius" | tee -a $out/progress.log
  rm -rf /tmp/prof_rad_$ctr
  SZG_RADIUS_HITS=500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_rad_$ctr -- python3 $GRAFT_REPO_ROOT/scripts/dev_one.py 12500000 384 4 1 10 16 > /tmp/pmc.log 2>&1
  lc=$(echo $ctr | tr A-Z a-z)
  one /tmp/prof_rad_$ctr "*counter_collection.csv" $out/${tag}_pmc_cfg5radius_$lc.csv
done

echo "[hl] headline only: launches of 16 sweeps under --kernel-trace --stats (AverageNs / 16 = one sweep)" | tee -a $out/progress.log
rm -rf /tmp/prof_hl
SZG_LAUNCHES=12 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hl -- python3 $GRAFT_REPO_ROOT/scripts/dev_one.py 1000000 768 32 1 10 16 > /tmp/hl.log 2>&1
one /tmp/prof_hl "*kernel_stats.csv" $out/${tag}_headline_16sweep_kernel_stats.csv

echo "[mq] shared sweeps: kernel stats + counters" | tee -a $out/progress.log
for b in 64 32 16 8 4; do
  rm -rf /tmp/prof_mq$b
  SZG_BITS=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mq$b -- python3 $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /tmp/mq.log 2>&1
  one /tmp/prof_mq$b "*kernel_stats.csv" $out/${tag}_mq_${b}bit_kernel_stats.csv
done
# 8-bit rows on the exact int8 sweep (what radius batches take; SZG_BF16_8BIT=0)
rm -rf /tmp/prof_mq8i
SZG_BF16_8BIT=0 SZG_BITS=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mq8i -- python3 $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /tmp/mq.log 2>&1
one /tmp/prof_mq8i "*kernel_stats.csv" $out/${tag}_mq_8bit_int8_kernel_stats.csv
# one dimension WITHOUT a shape kernel of its own (4-bit rows of 1 024 dims: the any-shape int8 sweep)
rm -rf /tmp/prof_mq4g
SZG_BITS=4 SZG_DIM=1024 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mq4g -- python3 $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /tmp/mq.log 2>&1
one /tmp/prof_mq4g "*kernel_stats.csv" $out/${tag}_mq_4bit_dim1024_kernel_stats.csv
SZG_BITS=8 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_bf16d8_kernel<6, 1, true" $out/${tag}_pmc_mq_bf16_8bit_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
SZG_BF16_8BIT=0 SZG_BITS=8 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_i8s_kernel<3, 1" $out/${tag}_pmc_mq_i8_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
SZG_BITS=4 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_i8s_kernel<3, 1" $out/${tag}_pmc_mq_i8_4bit_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
SZG_BITS=32 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_bf16s_kernel<6, 1, true" $out/${tag}_pmc_mq_bf16_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
SZG_BITS=64 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_bf16s_kernel<6, 1, true" $out/${tag}_pmc_mq_bf16_64bit_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
SZG_BITS=16 bash $GRAFT_REPO_ROOT/scripts/pmc.sh "mq_score_bf16d_kernel<6, 1, true" $out/${tag}_pmc_mq_bf16_16bit_sweep_summary.txt $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /dev/null 2>&1 || true
python3 $GRAFT_REPO_ROOT/scripts/make_traffic.py $out $tag > $out/traffic.json || true
echo done | tee -a $out/progress.log
