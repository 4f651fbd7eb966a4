#!/usr/bin/env python3
"""Where do the small device copies and element-wise torch kernels of a RepRFN inference frame come from?  (torch
profiler with Python stacks; prints the source lines inside superresolution_amd/ that launch aten::copy_ and friends.)"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from superresolution_amd.reprfn import RepRFN

torch.manual_seed(0)
m = RepRFN(3, 3, 48, 4).cuda().eval()
x = torch.rand(1, 3, 540, 960, device="cuda")
for _ in range(3):
    m.infer(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.infer(x)
    torch.cuda.synchronize()
ops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
print("aten ops of one frame:", ops.most_common(20))
where = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::mul", "aten::contiguous", "aten::clone", "aten::sum", "aten::to", "aten::_to_copy", "aten::index", "aten::pad", "aten::einsum") and ev.stack:
        frames = [f for f in ev.stack if "superresolution_amd" in f]
        where[(ev.name, frames[0] if frames else ev.stack[0])] += 1
for (name, frame), n in where.most_common(25):
    print(n, name, frame[-110:])
