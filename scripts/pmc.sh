#!/bin/bash
# PMC passes (rocprofv3 --kernel-trace --pmc <set>, one run per set, never with other trace
# domains) over one target command; per-launch averages of every counter for the kernels whose
# name contains $1.   bash scripts/pmc.sh <kernel substring> <summary file> <python script> [args...]
set -e
pat=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
: > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- python3 "$@" > /tmp/pmc_run.log 2>&1 || { echo "set $i failed: $set" >> $out; tail -3 /tmp/pmc_run.log >> $out; continue; }
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$pat" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
names = set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']:
        continue
    names.add(r['Kernel_Name'][:110])
    a = acc[r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print("%-32s per launch %.5g  (%d launches)" % (k, v / max(n, 1), n))
for n in sorted(names):
    print("   kernel:", n)
PY
done
cat $out
