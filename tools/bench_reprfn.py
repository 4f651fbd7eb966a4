#!/usr/bin/env python3
"""Times RepRFN (deploy form) at 540x960 -> 2160x3840 (dev helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import RepRFN
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(42)
m = RepRFN().cuda().eval()
for mod in m.modules():
    if hasattr(mod, "switch_to_deploy") and mod is not m:
        mod.switch_to_deploy()
x = torch.rand(1, 3, 540, 960, generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    y = m.infer(x); torch.cuda.synchronize()
    assert y.shape == (1, 3, 2160, 3840) and bool(torch.isfinite(y).all())
    t0 = time.perf_counter()
    for _ in range(iters):
        m.infer(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"RepRFN deploy 540x960 -> 2160x3840: {dt*1e3:.2f} ms/frame = {1/dt:.1f} frames/s", flush=True)
