#!/bin/bash
# per-kernel register / scratch summary of one .hip file (compiles to /dev/null): tools/kernel_res.sh april_asr_amd/csrc/kernels_gemm_pp.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I"$(dirname "$0")/../april_asr_amd/csrc" \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 | python3 -c '
import sys, re
cur = None
for line in sys.stdin:
    if "error" in line or "warning" in line: print(line.rstrip())
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; continue
    for key in ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "VGPRs Spill:", "TotalSGPRs:", "Occupancy [waves/SIMD]:"):
        if cur is not None and key in line: cur[key] = line.split(key)[1].split()[0]
    if cur is not None and "LDS Size" in line:
        print("%-90s sgpr %3s vgpr %3s agpr %3s scratch %4s spill %3s occ %s" % (cur["name"][:90], cur.get("TotalSGPRs:"), cur.get("VGPRs:"), cur.get("AGPRs:"), cur.get("ScratchSize [bytes/lane]:"), cur.get("VGPRs Spill:"), cur.get("Occupancy [waves/SIMD]:")))
        cur = None
'
