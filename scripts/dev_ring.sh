#!/bin/bash
set -e
for lib in syzgydb_amd/variants/libsyzgy_scan_prev.so syzgydb_amd/libsyzgy_scan.so; do
  echo "== $lib"
  SZG_LIB_PATH=$lib SZG_NQ=512 SZG_AB=blocks_per_cu:1,2,3,4 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -4 /tmp/o.txt
  SZG_LIB_PATH=$lib SZG_NQ=512 SZG_BITS=8 SZG_AB=blocks_per_cu:2,3,4 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -3 /tmp/o.txt
  SZG_LIB_PATH=$lib SZG_NQ=256 SZG_BITS=4 SZG_DIM=384 SZG_AB=blocks_per_cu:2,3,4 python scripts/dev_ab.py 12500032 > /tmp/o.txt; head -3 /tmp/o.txt
done
