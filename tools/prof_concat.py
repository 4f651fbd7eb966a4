#!/usr/bin/env python3
"""Run the two-source 32 + 64 -> 64 layer (`Cfg<6,2,2>`) at 2160x3840 a few times (for rocprofv3 counter collection)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
dev = torch.device("cuda")
H, W = 2160, 3840
xa = torch.relu(torch.rand(1, H, W, 32, device=dev) - 0.5).to(torch.bfloat16)
xb = torch.relu(torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16)
wp = F_.conv3x3_pack_weights(torch.randn(64, 96, 3, 3, device=dev) * 0.05, 32, 32, 64, 64)
b = torch.randn(64, device=dev) * 0.1
out = torch.empty((1, H, W, 64), dtype=torch.bfloat16, device=dev)
for _ in range(6):
    F_.conv3x3(xa, wp, b, 64, src_b=xb, act=1, out=out)
torch.cuda.synchronize()
