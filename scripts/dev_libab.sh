#!/bin/bash
# Same-box A/B of two library builds on the headline shape: boxes of the pool differ by ~4 %,
# so builds are only comparable within one gpurun call.
#   bash scripts/dev_libab.sh path/to/libA.so path/to/libB.so [rows...]
set -e
a=$1; b=$2; shift 2
rows=${@:-1000000}
for rep in 1 2; do
  for lib in $a $b; do
    echo "== $lib"
    SZG_LIB_PATH=$lib python scripts/dev_small.py $rows
  done
done
