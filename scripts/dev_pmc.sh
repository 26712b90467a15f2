#!/bin/bash
# PMC passes over one scan config (dev): usage dev_pmc.sh <tag> <rows> <dim> <bits> <metric> <k>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${tag}_$i -- python3 $GRAFT_REPO_ROOT/scripts/dev_one.py "$@" > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan_kernel' not in r['Kernel_Name']:
        continue
    a = acc[r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print("%-44s per launch %.4g  (%d records)" % (k, v / max(n, 1) , n))
PY
done
