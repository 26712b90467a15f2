#!/bin/bash
# every element width: tiled (SZG_TILES_ALL=1) vs the default layout, same box
set -e
for t in "" "1"; do
  if [ -n "$t" ]; then export SZG_TILES_ALL=1; echo "== tiled"; else unset SZG_TILES_ALL; echo "== default"; fi
  SZG_NQ=512 SZG_AB=blocks_per_cu:0,1,2,3 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -4 /tmp/o.txt
  SZG_DIM=384 SZG_NQ=512 SZG_AB=blocks_per_cu:0,2 python scripts/dev_ab.py 1000000 > /tmp/o.txt; head -2 /tmp/o.txt
  SZG_BITS=8 SZG_NQ=512 SZG_AB=blocks_per_cu:0,2,3 python scripts/dev_ab.py 4000000 > /tmp/o.txt; head -3 /tmp/o.txt
  SZG_BITS=16 SZG_NQ=512 SZG_AB=blocks_per_cu:0,2,3 python scripts/dev_ab.py 2000000 > /tmp/o.txt; head -3 /tmp/o.txt
  SZG_METRIC=0 SZG_K=101 SZG_NQ=256 SZG_AB=blocks_per_cu:0,2 python scripts/dev_ab.py 1250048 > /tmp/o.txt; head -2 /tmp/o.txt
done
