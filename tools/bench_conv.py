#!/usr/bin/env python3
"""Micro-benchmark of the 3x3 MFMA convolution at 4K (dev helper, run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_

H, W = 2160, 3840
dev = torch.device("cuda")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


_SPARSE = False


def case(name, ca, cb, cout, res=False, f32=False, act=1, alpha=1.0, h=H, w=W, n=1):
    xa = (torch.rand(n, h, w, ca, device=dev) - 0.5).to(torch.bfloat16)
    if _SPARSE:
        xa = torch.relu(xa)
    xb = (torch.rand(n, h, w, cb, device=dev) - 0.5).to(torch.bfloat16) if cb else None
    wt = torch.randn(cout, ca + cb, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev) * 0.1
    wp = F_.conv3x3_pack_weights(wt, ca, ca, cb, cb)
    r = (torch.rand(n, h, w, F_.pad16(cout), device=dev) - 0.5).to(torch.bfloat16) if res else None
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev) if f32 else torch.empty((n, h, w, F_.pad16(cout)), dtype=torch.bfloat16, device=dev)
    fn = lambda: F_.conv3x3(xa, wp, b, cout, src_b=xb, res=r, act=act, alpha=alpha, out_nchw_f32=f32, out=out)
    med, best = timeit(fn)
    flop = 2.0 * n * h * w * cout * (ca + cb) * 9
    byts = n * h * w * ((ca + cb) * 2 + (cout * 4 if f32 else F_.pad16(cout) * 2) + (F_.pad16(cout) * 2 if res else 0))
    print(f"{name:28s} median {med:7.3f} ms  best {best:7.3f} ms  {flop / med / 1e9:8.1f} TFLOP/s  {byts / med / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        case("64->64 relu", 64, 0, 64)
        case("64->64 relu 16x1024^2", 64, 0, 64, h=1024, w=1024, n=16)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "res":
        case("64->64 relu 16x1024^2", 64, 0, 64, h=1024, w=1024, n=16)
        case("64->64 +res 16x1024^2", 64, 0, 64, res=True, act=0, alpha=0.2, h=1024, w=1024, n=16)
        case("64->64 +res 4K", 64, 0, 64, res=True, act=0, alpha=0.2)
        case("16->64 relu 4K", 16, 0, 64)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sparse":
        # ReLU-like operands (half the values exactly zero): the chip draws less power than on dense random
        # data and the clocks stay up, as in the real network
        def sparse_case(name, **kw):
            global _SPARSE
            _SPARSE = True
            case(name, 64, 0, 64, **kw)
            _SPARSE = False
        sparse_case("64->64 relu 16x1024^2 sparse", h=1024, w=1024, n=16)
        sparse_case("64->64 relu 4K sparse")
        x = torch.relu(torch.rand(16, 1024, 1024, 64, device=dev) - 0.5).to(torch.bfloat16)
        dz = torch.relu(torch.rand(16, 1024, 1024, 64, device=dev) - 0.5).to(torch.bfloat16)
        dw = torch.empty(64, 64, 3, 3, device=dev); db = torch.empty(64, device=dev)
        med, best = timeit(lambda: F_.conv3x3_wgrad(x, dz, dw, db, 64, 64))
        print(f"wgrad 64x64 sparse median {med:7.3f} ms best {best:7.3f} ms", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bands":
        for hh in (2160, 1080, 540, 360, 216, 108, 54, 27):
            case(f"64->64 relu h={hh}", 64, 0, 64, h=hh)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        case("64->64 relu", 64, 0, 64)
        case("16->64 relu", 16, 0, 64)
        case("64->3 f32 nchw", 64, 0, 3, f32=True, act=0)
        sys.exit(0)
    case("64->64 relu", 64, 0, 64)
    case("64->64 0.2*conv+res", 64, 0, 64, res=True, act=0, alpha=0.2)
    case("(32+64)->64 relu", 32, 64, 64)
    case("16->64 relu", 16, 0, 64)
    case("64->3 f32 nchw", 64, 0, 3, f32=True, act=0)
    case("16->32 relu LR", 16, 0, 32, h=540, w=960)
    case("32->32 relu LR", 32, 0, 32, h=540, w=960)
    case("64->64 relu 16x1024^2", 64, 0, 64, h=1024, w=1024, n=16)


def wgrad_case(name, ci, co, h=1024, w=1024, n=16):
    x = (torch.rand(n, h, w, ci, device=dev) - 0.5).to(torch.bfloat16)
    dz = (torch.rand(n, h, w, co, device=dev) - 0.5).to(torch.bfloat16)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    db = torch.empty(co, device=dev)
    fn = lambda: F_.conv3x3_wgrad(x, dz, dw, db, ci, co)
    med, best = timeit(fn)
    flop = 2.0 * n * h * w * co * ci * 9
    byts = n * h * w * (ci + co) * 2
    print(f"wgrad {name:22s} median {med:7.3f} ms  best {best:7.3f} ms  {flop / med / 1e9:8.1f} TFLOP/s  {byts / med / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("quick", "one", "bands", "res", "sparse")):
    wgrad_case("64x64 16x1024^2", 64, 64)
    wgrad_case("32x64 16x1024^2", 32, 64)
    wgrad_case("16x64 16x1024^2", 16, 64)
    wgrad_case("64x16 16x1024^2", 64, 16)
    wgrad_case("32x32 16x256^2", 32, 32, h=256, w=256)
