#!/usr/bin/env python3
"""Repeatability of single kernels while ANOTHER process uses the same GPU (dev helper).
Each process repeats one launch on fixed inputs and counts results that differ from the first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
tag = sys.argv[1]
reps = int(sys.argv[2])
torch.manual_seed(1)
x = (torch.rand(2, 192, 320, 64, device="cuda") - 0.5).to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
b = torch.randn(64, device="cuda") * 0.1
wp = F_.conv3x3_pack_weights(w, 64, 64)
dw = torch.empty(64, 64, 3, 3, device="cuda"); db = torch.empty(64, device="cuda")
w1 = torch.randn(64, 64, 1, 1, device="cuda") * 0.1
fns = {
    "conv3x3": lambda: F_.conv3x3(x, wp, b, 64, act=1),
    "wgrad": lambda: (F_.conv3x3_wgrad(x, x, dw, db, 64, 64), dw.clone())[1],
    "direct1x1": lambda: F_.conv2d_direct(x, w1, b, 64, act=1),
    "relu_mask": lambda: F_.relu_mask(x, x),
}
for name, fn in fns.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    for _ in range(reps):
        y = fn()
        if not torch.equal(y, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"[{tag}] {name:10s}: {bad} of {reps} repetitions differ ({time.time() - t0:.1f} s)", flush=True)
