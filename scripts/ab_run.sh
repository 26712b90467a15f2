# A/B of library builds on one box (gpurun): every variant x every config, twice, interleaved.
cd $GRAFT_REPO_ROOT
CFG=${CFG:-"12500000:384:4:1:10 12500000:384:4:1:10:0.43 4000000:768:4:1:10 1000000:768:8:1:10 4000000:768:8:1:10 1000000:768:32:1:10 1250000:768:32:0:100"}
for rep in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset SZG_LIB_PATH; else export SZG_LIB_PATH=$GRAFT_REPO_ROOT/syzgydb_amd/variants/libsyzgy_scan_$v.so; fi
  timeout -k 5 200 python scripts/dev_cfg.py $CFG || exit 1
done
done
unset SZG_LIB_PATH
