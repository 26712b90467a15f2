#!/bin/bash
# A/B the ring-depth variants of the f32 scan (dev only)
set -e
for v in r4_b4 r6_b4; do
  echo "== $v"; SZG_LIB_PATH=syzgydb_amd/variants/libsyzgy_scan_$v.so SZG_NQ=1024 SZG_AB=blocks_per_cu:4,3,2 python scripts/dev_ab.py 1000000 | head -3
done
for v in r8_b2 r12_b2 r16_b2; do
  echo "== $v"; SZG_LIB_PATH=syzgydb_amd/variants/libsyzgy_scan_$v.so SZG_NQ=1024 SZG_AB=blocks_per_cu:2,1 python scripts/dev_ab.py 1000000 | head -2
  echo "== $v bt128"; SZG_LIB_PATH=syzgydb_amd/variants/libsyzgy_scan_$v.so SZG_OPTS=block_threads=128 SZG_NQ=1024 SZG_AB=blocks_per_cu:4,3,2 python scripts/dev_ab.py 1000000 | head -3
done
