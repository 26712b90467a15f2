#!/bin/bash
# PMC passes over the f32 shared sweep (dev)
set -e
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcmq_$i -- python3 $GRAFT_REPO_ROOT/scripts/dev_mq_one.py > /tmp/pmcmq.log 2>&1
  f=$(find /tmp/pmcmq_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'mq_score_kernel' not in r['Kernel_Name'] or 'true' not in r['Kernel_Name']:
        continue
    a = acc[r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print("%-44s per launch %.4g  (%d records)" % (k, v / max(n, 1) , n))
PY
done
