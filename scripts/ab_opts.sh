# A/B of tunables on ONE library build (gpurun): OPTSETS="a=1,b=2 c=3 -" (a "-" = defaults), twice.
cd $GRAFT_REPO_ROOT
CFG=${CFG:-"12500000:384:4:1:10 12500000:384:4:1:10:0.43 4000000:768:4:1:10 1000000:768:8:1:10 4000000:768:8:1:10 1000000:768:32:1:10 1000000:384:32:1:10 1250000:768:32:0:100 125056:768:32:1:11"}
for rep in 1 2; do
for o in ${OPTSETS:--}; do
  if [ "$o" = "-" ]; then oo=""; else oo=$o; fi
  SZG_OPTS=$oo timeout -k 5 200 python scripts/dev_cfg.py $CFG | sed "s/^default   /$(printf '%-10s' ${o:0:10})/" || exit 1
done
done
