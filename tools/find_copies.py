#!/usr/bin/env python3
"""Where do device-to-device copies come from in one RepRFN / temporal inference frame? (dev helper)"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "reprfn"
sites = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode

class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "contiguous", "_to_copy", "cat", "fill_", "zero_", "add", "mul", "abs")):
            st = [f for f in traceback.extract_stack() if "superresolution_amd" in f.filename]
            key = (name, tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]))
            sites[key] += 1
        return func(*args, **(kwargs or {}))

g = torch.Generator().manual_seed(1)
if which == "reprfn":
    from superresolution_amd import RepRFN
    m = RepRFN().cuda().eval()
    for mod in m.modules():
        if mod is not m and hasattr(mod, "switch_to_deploy"):
            mod.switch_to_deploy()
    x = torch.rand(1, 3, 540, 960, generator=g).cuda()
    run = lambda: m.infer(x)
else:
    from superresolution_amd.temporal import SuperSampler
    m = SuperSampler(3, 10, 4, 1).cuda().eval()
    h, w = 544, 960
    lr = torch.rand(1, 3, h, w, generator=g).cuda(); hb = torch.rand(1, 10, 4 * h, 4 * w, generator=g).cuda()
    lb = torch.rand(1, 10, h, w, generator=g).cuda(); mv = torch.randn(1, 1, 2, h, w, generator=g).cuda()
    hf = torch.rand(1, 1, 3, h, w, generator=g).cuda(); hbuf = torch.rand(1, 1, 10, h, w, generator=g).cuda()
    run = lambda: m.infer(lr, hb, lb, mv, hf, hbuf)
with torch.no_grad():
    run(); run()
    with Spy():
        run()
for (name, st), n in sites.most_common(40):
    print(n, name, " <- ".join(st))
