#!/usr/bin/env python3
"""Repeats forward+backward on fixed inputs and checks that every repetition gives finite, bit-identical
gradients (dev helper: hunts races / uninitialised reads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import SuperSampler

torch.manual_seed(42)
m = SuperSampler(3, 10, 1).cuda().train()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for shape in [(1, 32, 32), (2, 24, 40), (1, 64, 96), (3, 16, 136)]:
    b, sh, sw = shape
    g = torch.Generator().manual_seed(sh)
    lr = torch.rand(b, 1, 3, sh, sw, generator=g).cuda()
    hb = torch.rand(b, 1, 10, 4 * sh, 4 * sw, generator=g).cuda()
    ref = None
    for r in range(reps):
        m.zero_grad(set_to_none=True)
        junk = torch.full((64 * 1024 * 1024,), float("nan"), device="cuda")  # poison freed memory
        del junk
        out = m({"lr_frames": lr, "hr_buffers": hb})["sr_frames"]
        out.mean().backward()
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        if not bool(torch.isfinite(flat).all()):
            names = [n for n, p in m.named_parameters() if not bool(torch.isfinite(p.grad).all())]
            print(f"shape {shape} rep {r}: NON-FINITE grads in {names[:6]}", flush=True)
            bad += 1
        elif ref is None:
            ref = flat.clone()
        elif not torch.equal(ref, flat):
            d = (ref - flat).abs().max().item()
            print(f"shape {shape} rep {r}: grads differ from rep 0 by {d:.3e}", flush=True)
            bad += 1
    print(f"shape {shape}: done", flush=True)
print("bad repetitions:", bad)
sys.exit(1 if bad else 0)
