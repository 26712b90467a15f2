#!/bin/bash
set -e
run() { echo "== dim=$1 bits=$2 metric=$3 k=$4 rows=$5"; SZG_DIM=$1 SZG_BITS=$2 SZG_METRIC=$3 SZG_K=$4 SZG_NQ=256 SZG_AB=blocks_per_cu:0,2,3,4 python scripts/dev_ab.py $5 > /tmp/ab.out; head -4 /tmp/ab.out; }
run 768 8 1 11 4000000
run 768 4 1 11 8000000
run 768 16 1 11 2000000
run 384 8 1 11 8000000
