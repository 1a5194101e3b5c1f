#!/bin/bash
# kernel_regs.sh [-Dflags]: registers, spills and scratch of every kernel of elbo_mfma.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$(dirname "$0")/../ivae_amd/csrc/elbo_mfma.hip" -o /dev/null 2>&1 |
  python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS"):
        print("%-60s sgpr %3s vgpr %3s agpr %3s scratch %4s sspill %3s vspill %3s occ %s" % (cur["name"][:60], cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill"), cur.get("Occupancy [waves/SIMD]")))
'
