#!/usr/bin/env python3
"""Training step of RepRFN (train form, nine branches per block) on 16 LR crops of 256x256 -> HR 1024x1024 (dev helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import RepRFN
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 256
torch.manual_seed(42)
m = RepRFN().cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, S, S, generator=g).cuda()
t = torch.rand(B, 3, 4 * S, 4 * S, generator=g).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = (m(x) - t).abs().mean()
    loss.backward()
    opt.step()
    return loss
l0 = float(step().detach()); step(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert float(loss.detach()) < l0 and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
print(f"RepRFN train step, {B} crops 256^2 -> 1024^2: {dt*1e3:.1f} ms/step = {B/dt:.0f} crops/s (loss {l0:.4f} -> {float(loss.detach()):.4f})", flush=True)
