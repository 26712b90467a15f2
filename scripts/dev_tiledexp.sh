#!/bin/bash
# timing experiment: tiled (1 KB contiguous) addressing in the dense phase, L=4 maps (answers are wrong)
set -e
for lib in syzgydb_amd/libsyzgy_scan.so syzgydb_amd/variants/libsyzgy_scan_tiledexp.so; do
  echo "== $lib"
  SZG_LIB_PATH=$lib SZG_DIM=384 SZG_BITS=4 SZG_NQ=256 SZG_OPTS=tie_mode=1 SZG_AB=blocks_per_cu:2,3 python scripts/dev_ab.py 12500032 > /tmp/o.txt; head -2 /tmp/o.txt
  SZG_LIB_PATH=$lib SZG_DIM=768 SZG_BITS=8 SZG_NQ=256 SZG_OPTS=tie_mode=1,lanes_per_row=4 SZG_AB=blocks_per_cu:2,3 python scripts/dev_ab.py 4000000 > /tmp/o.txt; head -2 /tmp/o.txt
done
