#!/usr/bin/env python3
"""Run one conv case a few times (for rocprofv3 counter collection)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_
dev = torch.device("cuda")
H, W = 2160, 3840
ca = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
res = len(sys.argv) > 3 and sys.argv[3] == "res"
xa = (torch.rand(1, H, W, ca, device=dev) - 0.5).to(torch.bfloat16)
wt = torch.randn(cout, ca, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev) * 0.1
wp = F_.conv3x3_pack_weights(wt, ca, ca)
r = (torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16) if res else None
out = torch.empty((1, H, W, 64), dtype=torch.bfloat16, device=dev)
for _ in range(5):
    F_.conv3x3(xa, wp, b, cout, res=r, act=1, out=out)
torch.cuda.synchronize()
