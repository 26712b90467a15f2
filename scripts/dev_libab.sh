#!/bin/bash
# same-box A/B of two library builds on the headline shape (dev only)
set -e
for rep in 1 2; do
for lib in syzgydb_amd/variants/libsyzgy_scan_oldscan.so syzgydb_amd/libsyzgy_scan.so; do
  echo "== $lib"
  SZG_LIB_PATH=$lib python scripts/dev_small.py 1000000
done
done
