#!/bin/bash
# Register / scratch / LDS use of every kernel of one shared-sweep part (or a scan part):
#   scripts/kernel_resources.sh mq 2          -> kernels_mq.hip -DSZG_MQ_PART=2 (the 4-bit int8 sweeps)
#   scripts/kernel_resources.sh scan 32       -> kernels_scan.hip -DSZG_QBITS=32
# (hipcc -Rpass-analysis=kernel-resource-usage, device code only; no GPU needed)
set -euo pipefail
cd "$(dirname "$0")/../syzgydb_amd/csrc"
kind=${1:-mq}; part=${2:-2}
if [ "$kind" = mq ]; then src=kernels_mq.hip; def=-DSZG_MQ_PART=$part; else src=kernels_scan.hip; def=-DSZG_QBITS=$part; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $def ${VFLAGS:-} --cuda-device-only -c $src -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
name = None; row = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name = m.group(1); row = {}
    for key in ("VGPRs:", "AGPRs", "ScratchSize", "VGPRs Spill", "SGPRs Spill", "Occupancy", "LDS Size"):
        m = re.search(re.escape(key) + r".*?(\d+)", ln)
        if m and name: row[key] = int(m.group(1))
    if "LDS Size" in ln and name:
        import subprocess
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"szg::\(anonymous namespace\)::", "", dem).split("(")[0]
        print("%-64s vgpr %3d agpr %3d scratch %4d B  vgpr-spill %3d  occupancy %d" % (
            dem[:64], row.get("VGPRs:", -1), row.get("AGPRs", 0), row.get("ScratchSize", 0), row.get("VGPRs Spill", 0), row.get("Occupancy", 0)))
        name = None
'
