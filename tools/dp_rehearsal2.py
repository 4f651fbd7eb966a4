#!/usr/bin/env python3
"""Cold-start determinism check with two processes sharing one GPU (dev helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from superresolution_amd import SuperSampler
torch.manual_seed(42)
m = SuperSampler(3, 10, 1).cuda().train()
g = torch.Generator().manual_seed(7)
lr = torch.rand(1, 1, 3, 32, 32, generator=g).cuda(); hb = torch.rand(1, 1, 10, 128, 128, generator=g).cuda()
runs = []
outs = []
for r in range(3):
    m.zero_grad(set_to_none=True)
    out = m({"lr_frames": lr, "hr_buffers": hb})["sr_frames"]
    out.mean().backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.clone() for n, p in m.named_parameters()})
    outs.append(out.detach().clone())
for r in (0, 1):
    if not torch.equal(outs[r], outs[2]):
        print(f"[rank {rank}] run {r}: FORWARD output differs from run 2 by {(outs[r]-outs[2]).abs().max().item():.3e}", flush=True)
    diff = [(n, (runs[r][n] - runs[2][n]).abs().max().item()) for n in runs[2] if not torch.equal(runs[r][n], runs[2][n])]
    if diff:
        print(f"[rank {rank}] run {r}: {len(diff)} gradient tensors differ from run 2: {diff[:8]}", flush=True)
print(f"[rank {rank}] checked", flush=True)
dist.destroy_process_group()
