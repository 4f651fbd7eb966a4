#!/usr/bin/env python3
"""Run the head form of the pair kernel (64 -> 64 -> 3, fp32 NCHW out) at 2160x3840 a few times (for rocprofv3 counter
collection).  SRHIP_PAIR_HEAD=2 in the environment selects the two-waves-per-layer form."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
dev = torch.device("cuda")
H, W = 2160, 3840
x = torch.relu(torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16)
wp1 = F_.conv3x3_pack_weights(torch.randn(64, 64, 3, 3, device=dev) * 0.05, 64, 64)
wp2 = F_.conv3x3_pack_weights(torch.randn(3, 64, 3, 3, device=dev) * 0.05, 64, 64)
b1, b2 = torch.randn(64, device=dev) * 0.1, torch.randn(3, device=dev) * 0.1
for _ in range(6):
    F_.conv3x3_pair_head(x, wp1, b1, wp2, b2, 3)
torch.cuda.synchronize()
