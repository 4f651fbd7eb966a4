#!/usr/bin/env python3
"""RepRFN training step: host time to issue the step vs wall time with the GPU drained (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import RepRFN
from superresolution_amd.dp import FlatGradAllReducer

dev = torch.device("cuda")
torch.manual_seed(42)
model = RepRFN().to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
red = FlatGradAllReducer(params)
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)
if len(sys.argv) > 1 and sys.argv[1] == "cpu-data":      # as bench.py makes its batch: host generator, pageable upload
    g = torch.Generator().manual_seed(42)
    x = torch.rand(16, 3, 256, 256, generator=g).to(dev)
    target = torch.rand(16, 3, 1024, 1024, generator=g).to(dev)
else:
    x = torch.rand(16, 3, 256, 256, device=dev)
    target = torch.rand(16, 3, 1024, 1024, device=dev)


def step(parts):
    t = [time.perf_counter()]
    red.zero_grad(); t.append(time.perf_counter())
    out = model(x); t.append(time.perf_counter())
    loss = (out - target).abs().mean(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    red.all_reduce(); opt.step(); t.append(time.perf_counter())
    for i in range(5):
        parts[i] += t[i + 1] - t[i]


for _ in range(5):
    step([0] * 5)
torch.cuda.synchronize()
n = 20
parts = [0.0] * 5
t0 = time.perf_counter()
for _ in range(n):
    step(parts)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {t_issue / n * 1e3:.2f} ms/step, wall {t_all / n * 1e3:.2f} ms/step")
print("host parts (ms): zero_grad %.2f  forward %.2f  loss %.2f  backward %.2f  allreduce+adamw %.2f" % tuple(p / n * 1e3 for p in parts))
# same with a sync after each phase: GPU time per phase
parts = [0.0] * 5
for _ in range(n):
    t = [time.perf_counter()]
    red.zero_grad(); torch.cuda.synchronize(); t.append(time.perf_counter())
    out = model(x); torch.cuda.synchronize(); t.append(time.perf_counter())
    loss = (out - target).abs().mean(); torch.cuda.synchronize(); t.append(time.perf_counter())
    loss.backward(); torch.cuda.synchronize(); t.append(time.perf_counter())
    red.all_reduce(); opt.step(); torch.cuda.synchronize(); t.append(time.perf_counter())
    for i in range(5):
        parts[i] += t[i + 1] - t[i]
print("synced parts (ms): zero_grad %.2f  forward %.2f  loss %.2f  backward %.2f  allreduce+adamw %.2f" % tuple(p / n * 1e3 for p in parts))
