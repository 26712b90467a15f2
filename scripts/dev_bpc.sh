#!/bin/bash
# blocks_per_cu A/B across the configs (dev only)
set -e
run() { echo "== dim=$1 bits=$2 metric=$3 k=$4 rows=$5"; SZG_DIM=$1 SZG_BITS=$2 SZG_METRIC=$3 SZG_K=$4 SZG_NQ=512 SZG_AB=blocks_per_cu:1,2,3,4 python scripts/dev_ab.py $5 > /tmp/ab.out; head -4 /tmp/ab.out; }
run 384 32 1 11 1000000
run 768 32 0 11 1000000
run 768 32 0 101 1250048
run 768 64 0 11 500032
run 768 16 1 11 1000000
run 768 8 1 11 1000000
run 768 4 1 11 2000000
run 384 4 1 11 12500032
run 128 32 1 11 2000000
run 1536 32 1 11 500032
