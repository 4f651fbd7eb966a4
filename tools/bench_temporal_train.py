#!/usr/bin/env python3
"""Time one training step (forward + loss + backward + AdamW) of the temporal SuperSampler on the HIP path (dev helper,
run on the GPU box).  usage: bench_temporal_train.py [batch] [lr_size]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd.temporal import SuperSampler

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(42)
m = SuperSampler(3, 10, 4, 1).cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
g = torch.Generator().manual_seed(1)
args = [torch.rand(b, 3, s, s, generator=g), torch.rand(b, 10, 4 * s, 4 * s, generator=g), torch.rand(b, 10, s, s, generator=g),
        torch.randn(b, 1, 2, s, s, generator=g), torch.rand(b, 1, 3, s, s, generator=g), torch.rand(b, 1, 10, s, s, generator=g)]
args = [a.cuda() for a in args]
target = torch.rand(b, 3, 4 * s, 4 * s, generator=g).cuda()


def step():
    opt.zero_grad(set_to_none=True)
    sr, ew = m(*[a.clone() for a in args])
    w = torch.nn.functional.interpolate(ew, size=sr.shape[-2:], mode="nearest")
    loss = ((1.0 + w) * (sr - target).abs()).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"temporal training step: batch {b}, LR {s}x{s} -> HR {4 * s}x{4 * s}: {dt * 1e3:.1f} ms per step = {b / dt:.1f} crops/s, loss {float(loss.detach()):.4f}")
