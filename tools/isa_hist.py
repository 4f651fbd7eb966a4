#!/usr/bin/env python3
"""Instruction histogram of the steady-state row loop of the 64->64 3x3 kernel (dev helper).
Compiles the single-configuration variant 'base' of tools/conv_experiments.py to ISA."""
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_experiments as ce  # noqa: E402

os.makedirs(ce.BUILD, exist_ok=True)
name = sys.argv[1] if len(sys.argv) > 1 else "base"
base = open(os.path.join(ce.SRC, "conv3x3_mfma.hip")).read()
src = os.path.join(ce.BUILD, f"isa_{name}.hip")
open(src, "w").write(ce.patch(base, ce.ONLY_64 + ce.VARIANTS[name], name).replace('#include "srhip_common.h"', f'#include "{ce.SRC}/srhip_common.h"'))
out = os.path.join(ce.BUILD, f"isa_{name}.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
lines = lines[ends[0]:ends[1]]  # second kernel = conv3x3 (first is the pack kernel)
idx = [i for i, l in enumerate(lines) if "v_mfma" in l]
print("MFMAs in kernel:", len(idx))
seg = lines[idx[0]:idx[143] + 1]
ops = collections.Counter()
for l in seg:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    ops[l.split()[0]] += 1
print("instructions between the 1st and the 144th MFMA (two rows):", sum(ops.values()))
for k, v in ops.most_common(60):
    print(f"{k:32s}{v}")
print("ISA:", out)
