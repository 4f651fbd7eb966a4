#!/usr/bin/env python3
"""Which operator of the temporal network's forward stops repeating when another process shares the GPU?

`python3 tools/concurrency_check.py [procs] [steps]` starts `procs` copies of itself (default 2) that each build the
temporal SuperSampler at 544x960 -> 2176x3840, run the forward `steps` times and record a checksum of every tensor the
wrappers in superresolution_amd.functional return.  A step whose checksum list differs from the first step's names the
first operator (call index, function, output shape) that differed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(steps):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_
    from superresolution_amd.temporal import SuperSampler
    log = []

    def digest(t):
        v = t.contiguous().reshape(-1).view(torch.uint8)
        pad = (-v.numel()) % 8
        if pad:
            v = torch.cat((v, v.new_zeros(pad)))
        w = v.view(torch.int64)
        return int((w ^ (w >> 17)).sum().item()) & 0xFFFFFFFFFFFF

    def wrap(name, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    # a copy now (stream-ordered, no host wait inside the forward), its checksum after the forward
                    log.append((name, i, tuple(o.shape), o.detach().clone()))
            return out
        inner.__wrapped__ = fn
        return inner

    for name in dir(F_):
        fn = getattr(F_, name)
        if callable(fn) and getattr(fn, "__module__", "") == F_.__name__ and not name.startswith("_") and name not in ("check", "pad16", "invalidate_pack_cache"):
            setattr(F_, name, wrap(name, fn))

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(42)
    model = SuperSampler(3, 10, 4, 1).to(dev).eval()
    h, w = 544, 960
    lr = torch.rand(1, 3, h, w, generator=g).to(dev)
    hb = torch.rand(1, 10, 4 * h, 4 * w, generator=g).to(dev)
    lb = torch.rand(1, 10, h, w, generator=g).to(dev)
    mv = torch.randn(1, 1, 2, h, w, generator=g).to(dev)
    hf = torch.rand(1, 1, 3, h, w, generator=g).to(dev)
    hbuf = torch.rand(1, 1, 10, h, w, generator=g).to(dev)
    d_hb, d_lb, d_hbuf = hb[:, -1].clone(), lb[:, -1].clone(), hbuf[:, :, -1].clone()
    first = None
    bad = 0
    for s in range(steps):
        hb[:, -1].copy_(d_hb)
        lb[:, -1].copy_(d_lb)
        hbuf[:, :, -1].copy_(d_hbuf)
        del log[:]
        with torch.no_grad():
            out = model.infer(lr, hb, lb, mv, hf, hbuf)[0]
        log.append(("output", 0, tuple(out.shape), out))
        log.append(("input planes 0-8", 0, (9,), hb[:, :9].clone()))
        log.append(("fp32 gradient map of planes 0-2 after the forward", 0, (3,), F_.gradient_map.__wrapped__(hb[:, :3].contiguous(), 1e-6) if hasattr(F_.gradient_map, "__wrapped__") else hb[:, :1].clone()))
        if s > 0 and first is not None and os.environ.get("CHECK_WHERE"):
            for k, (a, b) in enumerate(zip(first_t, log)):
                if not torch.equal(a[3], b[3]):
                    d = (a[3] != b[3]).nonzero()
                    i0 = tuple(d[0].tolist())
                    if d.shape[1] == 4:
                        from collections import Counter
                        print("  x % 64:", sorted(Counter((d[:, 2] % 64).tolist()).items()), flush=True)
                        print("  channel:", sorted(Counter(d[:, 3].tolist()).items()), flush=True)
                        print("  (y, x // 64):", sorted(Counter(zip(d[:, 1].tolist(), (d[:, 2] // 64).tolist())).items())[:40], flush=True)
                    print(f"pid {os.getpid()} step {s} record {k} {b[0]} {b[2]}: {d.shape[0]} elements differ, first {d[0].tolist()} last {d[-1].tolist()}; "
                          f"values {float(a[3][i0])!r} vs {float(b[3][i0])!r}; max abs diff {float((a[3].float() - b[3].float()).abs().max())!r}", flush=True)
                    break
        if s == 1:
            first_t = list(log)
        log[:] = [(a, b, c, digest(t)) for a, b, c, t in log]
        if s == 0:      # warm-up: the first forward also packs the weights
            continue
        if first is None:
            first = list(log)
            continue
        if len(log) != len(first):
            print(f"pid {os.getpid()} step {s}: {len(log)} records instead of {len(first)}", flush=True)
            bad += 1
            continue
        for k, (a, b) in enumerate(zip(first, log)):
            if a != b:
                print(f"pid {os.getpid()} step {s}: first difference at record {k} of {len(log)}: {b[0]}[{b[1]}] {b[2]}"
                      f" (previous record: {log[k - 1][:3] if k else None})", flush=True)
                bad += 1
                break
    print(f"pid {os.getpid()}: {steps} steps, {bad} differing", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        sys.exit(child(int(sys.argv[2])))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(steps)]) for _ in range(procs)]
    rc = 0
    for p in ps:
        rc |= p.wait()
    sys.exit(rc)
