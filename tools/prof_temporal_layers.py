#!/usr/bin/env python3
"""Per-launch times of the temporal SuperSampler's (or RepRFN's) convolutions at the bench shape: every call of the
convolution entry points is bracketed by events; prints one line per call site with its shape, time and MFMA rate.
usage: prof_temporal_layers.py [temporal|reprfn] [h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_

which = sys.argv[1] if len(sys.argv) > 1 else "temporal"
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ((544, 960) if which == "temporal" else (540, 960))
calls = []
REC = [False]


def wrap(name, flops):
    orig = getattr(F_, name)

    def f(*a, **k):
        if not REC[0]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        calls.append((name, flops(a, k, out), e0, e1))
        return out
    setattr(F_, name, f)


def gemm_info(a, k, out):
    srcs = a[0]
    srcs = list(srcs) if isinstance(srcs, (tuple, list)) and not (len(srcs) == 3 and isinstance(srcs[1], int)) else [srcs]
    cin = sum(s[2] if isinstance(s, tuple) else s.shape[-1] for s in srcs)
    t0 = srcs[0][0] if isinstance(srcs[0], tuple) else srcs[0]
    ks, cout = a[4], a[3]
    o = out
    fl = 2.0 * o.shape[0] * o.shape[1] * o.shape[2] * cout * cin * ks * ks
    return f"k{ks} s{k.get('stride', 1)} {cin:4d}->{cout:4d} in {t0.shape[1]}x{t0.shape[2]} out {o.shape[1]}x{o.shape[2]} nsrc {len(srcs)} res {k.get('res') is not None}", fl


def c3_info(a, k, out):
    x = a[0]
    o = out[0] if isinstance(out, tuple) else out
    cin = x.shape[-1] + (k["src_b"].shape[-1] if k.get("src_b") is not None else 0)
    if k.get("ca_slice"):
        cin = k["ca_slice"][1]
    cout = a[3]
    fl = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * cin * 9
    return f"3x3 {cin:4d}->{cout:4d} {x.shape[1]}x{x.shape[2]}", fl


wrap("conv_gemm", gemm_info)
wrap("conv3x3", c3_info)
for nm in ("conv_transpose2x2s2", "attention_fuse", "resize_bilinear_nhwc", "pixel_shuffle4_conv3x3_c3", "pixel_unshuffle_to_nhwc_bf16",
           "gradient_map_unshuffle_nhwc", "max_pool2d_nhwc", "sigmoid_gate", "layernorm_cf", "conv2d_direct", "pixel_shuffle_nhwc_to_nchw_f32",
           "nchw_f32_to_nhwc_bf16", "esa_tail", "conv1x1_chain", "backward_warp", "backward_warp_nhwc", "dilate_velocity", "clamp_depth", "warp_concat_nhwc", "disocclusion", "concat_nhwc", "add_bf16"):
    if hasattr(F_, nm):
        wrap(nm, lambda a, k, out: ("", 0.0))

torch.manual_seed(42)
g = torch.Generator().manual_seed(3)
if which == "temporal":
    from superresolution_amd.temporal import SuperSampler
    m = SuperSampler(3, 10, 4, 1).cuda().eval()
    lr = torch.rand(1, 3, h, w, generator=g).cuda()
    hb = torch.rand(1, 10, 4 * h, 4 * w, generator=g).cuda()
    lb = torch.rand(1, 10, h, w, generator=g).cuda()
    mv = (torch.randn(1, 1, 2, h, w, generator=g)).cuda()
    hf = torch.rand(1, 1, 3, h, w, generator=g).cuda()
    hbuf = torch.rand(1, 1, 10, h, w, generator=g).cuda()
    run = lambda: m.infer(lr, hb.clone(), lb.clone(), mv, hf, hbuf.clone())
else:
    from superresolution_amd.reprfn import RepRFN
    m = RepRFN(3, 3, 48, 4, deploy=True).cuda().eval()
    lr = torch.rand(1, 3, h, w, generator=g).cuda()
    run = lambda: m.infer(lr) if hasattr(m, "infer") else m(lr)
with torch.no_grad():
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    NREP = 5
    per = None
    for _ in range(NREP):
        calls.clear()
        REC[0] = True
        run()
        REC[0] = False
        torch.cuda.synchronize()
        ts = [e0.elapsed_time(e1) * 1e3 for (_, _, e0, e1) in calls]
        per = ts if per is None else [min(a, b) for a, b in zip(per, ts)]
tot = 0.0
agg = {}
for (name, (desc, fl), _, _), t in zip(calls, per):
    tot += t
    key = (name, desc)
    c = agg.setdefault(key, [0, 0.0, fl])
    c[0] += 1
    c[1] += t
print(f"{'calls':>5} {'us total':>9} {'us each':>8} {'TFLOP/s':>8}  op")
for (name, desc), (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    rate = fl * n / (t * 1e-6) / 1e12 if fl else 0.0
    print(f"{n:5d} {t:9.1f} {t / n:8.1f} {rate:8.1f}  {name} {desc}")
print(f"sum of bracketed launches: {tot / 1e3:.3f} ms over {len(calls)} calls")
