#!/bin/bash
# shape-specialised vs any-shape scan kernels across the configs (dev only)
set -e
run() { echo "== dim=$1 bits=$2 metric=$3 k=$4 rows=$5"; SZG_DIM=$1 SZG_BITS=$2 SZG_METRIC=$3 SZG_K=$4 SZG_NQ=512 SZG_AB=shape_kernels:0,1 python scripts/dev_ab.py $5 > /tmp/ab.out; cat /tmp/ab.out; }
run 768 32 1 11 1000000
run 384 32 1 11 1000000
run 768 8 1 11 1000000
run 384 4 1 11 12500032
run 768 4 1 11 2000000
run 768 16 1 11 1000000
run 768 32 0 101 1250048
