# A/B of library builds on the shared (multi-query) sweeps, one box.  VARIANTS="default x y", BITS="8 4"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset SZG_LIB_PATH; else export SZG_LIB_PATH=$GRAFT_REPO_ROOT/syzgydb_amd/variants/libsyzgy_scan_$v.so; fi
  for b in ${BITS:-8}; do for d in ${DIMS:-768}; do
    SZG_BITS=$b SZG_DIM=$d timeout -k 5 200 python scripts/dev_mqab.py | tail -1 || exit 1
  done; done
done
done
unset SZG_LIB_PATH
