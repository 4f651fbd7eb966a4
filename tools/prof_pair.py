#!/usr/bin/env python3
"""Run the fused residual-pair kernel a few times at 2160x3840 (for rocprofv3 counter collection).
usage: prof_pair.py [mid]      mid: also write the intermediate map (training form)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
dev = torch.device("cuda")
H, W = 2160, 3840
keep = len(sys.argv) > 1 and sys.argv[1] == "mid"
x = (torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16)
w1 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b1 = torch.randn(64, device=dev) * 0.1
b2 = torch.randn(64, device=dev) * 0.1
wp1, wp2 = F_.conv3x3_pack_weights(w1, 64, 64), F_.conv3x3_pack_weights(w2, 64, 64)
out = torch.empty_like(x)
for _ in range(5):
    F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2, keep_mid=keep, out=out)
torch.cuda.synchronize()
