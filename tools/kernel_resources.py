#!/usr/bin/env python3
"""Register, LDS and occupancy figures of every kernel of the HIP translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include", "-c", "-o", "/dev/null",
       f"{ROOT}/lrr_amd/csrc/lrr_icp.hip", "-Rpass-analysis=kernel-resource-usage"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'waves/SIMD':>10s}")
for k, r in rows.items():
    if flt in k:
        print(f"{k[:44]:44s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('ScratchSize', 0):8d} {r.get('LDS', 0):7d} {r.get('Occupancy', 0):10d}")
