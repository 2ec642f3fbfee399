#!/usr/bin/env python
"""Registers / scratch / LDS / occupancy of every kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line each:

    python tools/kernel_resources.py onepose_amd/csrc/gatsspg_gemm_kernels.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

src = os.path.abspath(sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", src] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "TotalSGPRs"):
        m = re.search(r"\s" + k + r"[^:]*: (\d+)", line)
        if m and "Spill" not in line:
            cur[k.split(" ")[0]] = int(m.group(1))
if not rows:
    sys.stderr.write(err[-3000:])
    sys.exit(1)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("gatsspg::", "").split("(")[0][:110]
    print(f"{n:110s} vgpr {r.get('VGPRs'):3d} agpr {r.get('AGPRs'):3d} sgpr {r.get('TotalSGPRs'):3d} scratch {r.get('ScratchSize'):4d} "
          f"waves/SIMD {r.get('Occupancy')} lds {r.get('LDS')}")
