#!/usr/bin/env python3
"""Forward passes of the default network on fixed inputs while another process shares the GPU: which layer's
output first deviates from the first pass (dev helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import SuperSampler
from superresolution_amd import functional as F_
tag, reps = sys.argv[1], int(sys.argv[2])
torch.manual_seed(42)
m = SuperSampler(3, 10, 1).cuda().eval()
g = torch.Generator().manual_seed(7)
lr = torch.rand(1, 3, 32, 32, generator=g).cuda(); hb = torch.rand(1, 10, 128, 128, generator=g).cuda()
trace = []
real_conv, real_up = F_.conv3x3, F_.upsample_grid_bilinear_nhwc
def conv(*a, **k):
    y = real_conv(*a, **k); trace.append(("conv%d" % len(trace), y)); return y
def up(*a, **k):
    y = real_up(*a, **k); trace.append(("up%d" % len(trace), y)); return y
F_.conv3x3, F_.upsample_grid_bilinear_nhwc = conv, up
m.infer(lr, hb); torch.cuda.synchronize()
ref = [(n, t.clone()) for n, t in trace]
stats = {}
t0 = time.time()
for r in range(reps):
    trace.clear()
    m.infer(lr, hb)
    for (n, t), (_, rt) in zip(trace, ref):
        if not torch.equal(t, rt):
            bad = (t.float() - rt.float()).abs()
            nanc = int(torch.isnan(t.float()).sum())
            idx = torch.nonzero(bad.reshape(-1) > 0)[:3].flatten().tolist() if nanc == 0 else torch.nonzero(torch.isnan(t.float()).reshape(-1))[:3].flatten().tolist()
            stats.setdefault(n, []).append((r, float(bad.nan_to_num(1e9).max()), nanc, tuple(t.shape), idx))
            break
print(f"[{tag}] {reps} passes in {time.time() - t0:.1f} s; first-deviating layers: " + (str({k: (len(v), v[:2]) for k, v in stats.items()}) if stats else "none"), flush=True)
