#!/usr/bin/env python3
"""aten operators of one steady-state forward of the temporal SuperSampler (what torch still launches around libsrhip)."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from superresolution_amd.temporal import SuperSampler

torch.manual_seed(0)
m = SuperSampler(3, 10, 4, 1).cuda().eval()
h, w = 544, 960
mk = lambda *s: torch.rand(*s, device="cuda")
args = lambda: (mk(1, 3, h, w), mk(1, 10, 4 * h, 4 * w), mk(1, 10, h, w), torch.randn(1, 1, 2, h, w, device="cuda"), mk(1, 1, 3, h, w), mk(1, 1, 10, h, w))
for _ in range(3):
    m.infer(*args())
torch.cuda.synchronize()
a = args()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    m.infer(*a)
    torch.cuda.synchronize()
ops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
print("aten ops of one frame:", ops.most_common(30))
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::abs", "aten::max", "aten::lt", "aten::cat"):
        print(ev.name, [tuple(s) for s in (ev.input_shapes or [])][:2], round(ev.device_time_total, 1) if hasattr(ev, "device_time_total") else "")
