#!/usr/bin/env python3
"""Run the training step's two dominant kernels at its shape (16 x 1024 x 1024, 64 -> 64: the single-layer convolution
that every data gradient uses, and the weight-gradient kernel) and the recurrent network's fused trunk-input pass at
4K, three launches each (for rocprofv3 counter collection)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
dev = torch.device("cuda")
n, h, w = 16, 1024, 1024
x = torch.relu(torch.rand(n, h, w, 64, device=dev) - 0.5).to(torch.bfloat16)
dz = ((torch.rand(n, h, w, 64, device=dev) - 0.5) * (torch.rand(n, h, w, 64, device=dev) > 0.5)).to(torch.bfloat16)
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
wp = F_.conv3x3_pack_weights(wt, 64, 64)
out = torch.empty_like(x)
dw, db = torch.empty(64, 64, 3, 3, device=dev), torch.empty(64, device=dev)
H, W = 2160, 3840
gb, fr = torch.rand(1, 10, H, W, device=dev), torch.rand(1, 3, H, W, device=dev)
mv = torch.nn.functional.interpolate(torch.randn(1, 2, 28, 48, device=dev) * 4, size=(H, W), mode="bilinear", align_corners=True).clamp(-32, 32).contiguous()
for _ in range(3):
    F_.conv3x3(x, wp, b, 64, act=1, out=out)
    F_.conv3x3_wgrad(x, dz, dw, db, 64, 64)
    F_.warp_concat_nhwc(gb, fr, mv, "bilinear", True)
torch.cuda.synchronize()
