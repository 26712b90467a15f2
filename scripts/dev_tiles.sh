#!/bin/bash
# 4-bit rows: tiled (default) vs linear (SZG_NO_TILES=1) resident layout, same box
set -e
for t in "" "1"; do
  if [ -n "$t" ]; then export SZG_NO_TILES=1; echo "== linear"; else unset SZG_NO_TILES; echo "== tiled"; fi
  SZG_DIM=384 SZG_BITS=4 SZG_NQ=256 SZG_AB=blocks_per_cu:0,2,3,4 python scripts/dev_ab.py 12500032 | head -4
  SZG_DIM=768 SZG_BITS=4 SZG_NQ=256 SZG_AB=blocks_per_cu:0,2,3 python scripts/dev_ab.py 8000000 | head -3
  SZG_BITS=4 python scripts/dev_mqab.py | tail -1
  SZG_BITS=4 SZG_DIM=384 python scripts/dev_mqab.py | tail -1
  python scripts/dev_radius.py 2>/dev/null | tail -2
done
