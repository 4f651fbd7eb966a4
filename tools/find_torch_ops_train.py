#!/usr/bin/env python3
"""Which ATen operators run inside one temporal-network training step, from where, on how many bytes? (dev helper:
the step should be the library's kernels; every large torch elementwise pass is a candidate for fusion)"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from superresolution_amd.temporal import SuperSampler
from superresolution_amd.loss import enhance_weighted_l1

MIN_BYTES = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 20)   # smallest tensor that makes an operator count
sites = collections.Counter()
nbytes = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        big = 0
        flat = []
        for t in list(args) + [out]:
            flat.extend(t if isinstance(t, (list, tuple)) else [t])
        for t in flat:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                big = max(big, t.numel() * t.element_size())
        if big >= MIN_BYTES and not any(k in name for k in ("view", "reshape", "detach", "alias", "as_strided", "permute", "slice", "select", "transpose", "unsqueeze", "expand", "_unsafe_view", "t.default", "empty", "split", "unbind", "squeeze", "narrow")):
            st = [f for f in traceback.extract_stack() if "superresolution_amd" in f.filename]
            key = (name, tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]))
            sites[key] += 1
            nbytes[key] += big
        return out


dev = torch.device("cuda")
WHICH = os.environ.get("MODEL", "temporal")   # temporal | reprfn
B, S, SCALE = 16, 256, 4
torch.manual_seed(42)
g = torch.Generator().manual_seed(42)
if WHICH == "reprfn":
    from superresolution_amd import RepRFN
    model = RepRFN().to(dev).train()
    x = torch.rand(B, 3, S // SCALE, S // SCALE, generator=g).to(dev)
    target = torch.rand(B, 3, S, S, generator=g).to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = (model(x) - target).abs().mean()
        loss.backward()
        return loss
else:
    model = SuperSampler(3, 10, SCALE, 1).to(dev).train()
    ins = [torch.rand(B, 3, S, S, generator=g), torch.rand(B, 10, S * SCALE, S * SCALE, generator=g), torch.rand(B, 10, S, S, generator=g),
           torch.randn(B, 1, 2, S, S, generator=g), torch.rand(B, 1, 3, S, S, generator=g), torch.rand(B, 1, 10, S, S, generator=g)]
    ins = [t.to(dev) for t in ins]
    target = torch.rand(B, 3, S * SCALE, S * SCALE, generator=g).to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        sr, ew = model(*ins)
        loss = enhance_weighted_l1(sr, target, ew)
        loss.backward()
        return loss


if os.environ.get("FULL", "0") == "1":   # the whole training step of bench.py: bucketed reducer and fused AdamW around it
    from superresolution_amd.dp import FlatGradAllReducer
    params = [p for p in model.parameters() if p.requires_grad]
    red = FlatGradAllReducer(params, gather=True, buckets=2 if WHICH == "reprfn" else 4).attach_hooks()
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)
    inner = step

    def step():
        red.zero_grad()
        if WHICH == "reprfn":
            loss = (model(x) - target).abs().mean()
        else:
            sr, ew = model(*ins)
            loss = enhance_weighted_l1(sr, target, ew)
        loss.backward()
        red.all_reduce()
        opt.step()
        return loss

step(); step()
with Spy():
    step()
torch.cuda.synchronize()
for key, n in sorted(sites.items(), key=lambda kv: -nbytes[kv[0]])[:int(os.environ.get('TOP', '60'))]:
    print(f"{nbytes[key] / 1e6:9.1f} MB {n:3d}x {key[0]:40s} " + " <- ".join(key[1]))
print(sum(sites.values()), "operators")
