#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print VGPR / AGPR / scratch per kernel (dev helper)."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                      src, "-o", "/root/repo/build/_res.s", "-Rpass-analysis=kernel-resource-usage"] + extra,
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
for r in rows:
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["name"])
    name = name.replace("INS_3Cfg", "<").replace("EEEEEvNS_10ConvParamsE", ">")
    print(f"{name[:60]:60s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} scratch={r.get('ScratchSize')} "
          f"sgpr_spill={r.get('SGPRs Spill')} vgpr_spill={r.get('VGPRs Spill')} occ={r.get('Occupancy')}")
