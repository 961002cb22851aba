"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), device pass only.
Usage: python tools/kernel_resources.py zklaim_amd/csrc/msm.hip"""
import os, re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/tmp/_kres.o"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "LDS")]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    vals = []
    for k, short in keys:
        m = re.search(k + r": (\d+)", b)
        vals.append(f"{short} {m.group(1) if m else '?'}")
    print(f"{name[:100]:100s} " + "  ".join(vals))
