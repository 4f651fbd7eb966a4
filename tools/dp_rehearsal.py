#!/usr/bin/env python3
"""Two ranks sharing one GPU over gloo: stage-by-stage check of the collectives the bench uses (dev helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
def say(msg):
    print(f"[rank {rank} {time.time() % 1000:.1f}] {msg}", flush=True)

torch.cuda.set_device(0)
dist.init_process_group("gloo")
say("init done")
t = torch.ones(4, device="cuda") * (rank + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
say(f"cuda all_reduce ok {t.tolist()}")
dist.broadcast(t, src=0)
say("cuda broadcast ok")
dist.barrier()
say("barrier ok")
from superresolution_amd import SuperSampler
from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
torch.manual_seed(42 + rank)
m = SuperSampler(3, 10, 1).cuda().train()
broadcast_parameters(m, src=0)
say("broadcast_parameters ok")
red = FlatGradAllReducer(m.parameters())
lr = torch.rand(1, 1, 3, 32, 32, device="cuda"); hb = torch.rand(1, 1, 10, 128, 128, device="cuda")
out = m({"lr_frames": lr, "hr_buffers": hb})["sr_frames"]
say("forward ok")
red.zero_grad()
out.mean().backward()
torch.cuda.synchronize()
say(f"backward ok, |g| before all_reduce = {red.flat.abs().sum().item():.4e}, out finite {bool(torch.isfinite(out).all())}, nan params: "
    + str([n for n, p in m.named_parameters() if not bool(torch.isfinite(p.grad).all())][:4]))
red.all_reduce()
torch.cuda.synchronize()
say(f"grad all_reduce ok, |g| = {red.flat.abs().sum().item():.4e}")
dist.destroy_process_group()
say("done")
