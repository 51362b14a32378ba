#!/usr/bin/env python3
"""Register / scratch / occupancy table of the GEMM kernel instantiations (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_regs.py [file.hip]   (run from the repository root; cross-compiles for gfx950, no GPU needed)"""
import re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "april_asr_amd/csrc/kernels_gemm.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur)
    for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs", "LDS Size"):
        m = re.search(re.escape(k) + r"[^:\d]*: (\d+)", l)
        if m and cur is not None and k not in cur:
            cur[k] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
EPI = ["PARTIAL", "LSTM", "DSWISH", "HR", "RESID_SSQ", "SLOT_STORE", "XPART"]; AOP = ["-", "tanh", "scale"]
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    m = re.search(r"gemm_f32_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        mt, nt, e, a, wt, md, hand, nw = map(int, m.groups())
        n = "gemm %dx%d %-10s %-5s %s %s %s %dw" % (16 * mt, 16 * nt, EPI[e], AOP[a], "f16" if wt else "f32", "FULLK" if md else "slab ", "asm" if hand else "   ", nw)
    m = re.search(r"gemm_f32_zkernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        mt, nt, e, a, wt, md = map(int, m.groups())
        n = "gemm-z %dx%d %-10s %-5s %s %s" % (16 * mt, 16 * nt, EPI[e], AOP[a], "f16" if wt else "f32", "FULLK" if md else "slab ")
    print("%-60s vgpr %3s agpr %3s scratch %4s occ %s sgpr %s" % (n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRs")))
