#!/bin/bash
set -e
for v in r4 r6 r10 r12; do
  lib=syzgydb_amd/variants/libsyzgy_scan_$v.so
  echo "== $v bt256"; SZG_LIB_PATH=$lib SZG_NQ=512 SZG_AB=blocks_per_cu:1,2 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -2 /tmp/o.txt
  echo "== $v bt128"; SZG_LIB_PATH=$lib SZG_OPTS=block_threads=128 SZG_NQ=512 SZG_AB=blocks_per_cu:1,2,3 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -3 /tmp/o.txt
done
echo "== r8 (default lib) bt128"; SZG_OPTS=block_threads=128 SZG_NQ=512 SZG_AB=blocks_per_cu:1,2,3 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -3 /tmp/o.txt
echo "== r8 (default lib) bt256"; SZG_NQ=512 SZG_AB=blocks_per_cu:1,2 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -2 /tmp/o.txt
