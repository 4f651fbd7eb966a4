#!/usr/bin/env python3
"""A few launches of one of the kernels added last (for rocprofv3 counter passes):
    rocprofv3 --pmc <counters> -- python3 tools/prof_new_kernels.py <case>
cases: chain (160-160-160-64 1x1 chain at 544x960), chain_res (112-64-64-64 + residual), esa (ESA tail at 540x960 x 48),
loss (fused clip + L1 + SSIM forward and backward at 16 x 3 x 1024 x 1024), upbwd (x4 up-sampling backward at 16 x 1024 x 1024 x 32),
ln (channels-first LayerNorm forward + backward at 540 x 960 x 64 fp32), scaleadd (a + scale * y and its backward at 16 x 256 x 256 x 64), warp16 (bf16 NHWC warp of a 4K 16-channel frame, smooth motion),
pairbwd (both data gradients of an SRResidualBlock at 16 x 1024 x 1024), pairbits (the forward launch that emits the bits)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_

dev = torch.device("cuda")
case = sys.argv[1] if len(sys.argv) > 1 else "chain"
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5)
REP = 6
if case in ("chain", "chain_res"):
    ch = (160, 160, 160, 64) if case == "chain" else (112, 64, 64, 64)
    x = rnd(1, 544, 960, ch[0]).to(torch.bfloat16)
    layers = [((rnd(ch[l + 1], ch[l], 1, 1) * 0.2).float(), rnd(ch[l + 1]) * 0.1, F_.ACT_RELU) for l in range(3)]
    res = rnd(1, 544, 960, 64).to(torch.bfloat16) if case == "chain_res" else None
    for _ in range(REP):
        F_.conv1x1_chain(x, layers, alpha=0.45 if res is not None else 1.0, res=res)
elif case == "loss":
    from superresolution_amd.loss import FusedLossGroup
    sr = (torch.rand(16, 3, 1024, 1024, device=dev, generator=g) * 1.2 - 0.1).requires_grad_(True)
    hr = torch.rand(16, 3, 1024, 1024, device=dev, generator=g)
    lg = FusedLossGroup(1.0, 0.5, clip=(0.0, 10.0))
    for _ in range(REP):
        loss, _ = lg({"hr_frames": hr, "sr_frames": sr})
elif case == "upbwd":
    dy = rnd(16, 1024, 1024, 32).to(torch.bfloat16)
    y = torch.relu(rnd(16, 256, 256, 32)).to(torch.bfloat16)
    for _ in range(REP):
        F_.upsample_grid_bilinear_bwd_nhwc(dy, 256, 256, relu_out=y)
elif case == "ln":
    x, w, b = rnd(1, 64, 540, 960), rnd(64) + 1.0, rnd(64)
    dy = rnd(1, 64, 540, 960)
    for _ in range(REP):
        F_.layernorm_cf(x, w, b, 1e-6)
        F_.layernorm_cf_bwd(x, w, dy)
elif case == "scaleadd":
    a, y, d = (rnd(16, 256, 256, 64).to(torch.bfloat16) for _ in range(3))
    p = torch.tensor(0.45, device=dev)
    for _ in range(REP):
        F_.scale_add_bf16(a, y, p)
        F_.scale_add_bwd_bf16(d, y, p)
elif case == "warp16":
    fr = rnd(1, 2160, 3840, 16).to(torch.bfloat16)
    mv = torch.nn.functional.interpolate(torch.randn(1, 2, 28, 48, device=dev) * 4, size=(2160, 3840), mode="bilinear", align_corners=True)
    for _ in range(REP):
        F_.backward_warp_nhwc(fr, mv, "bilinear", True)
elif case == "esa":
    x, c1, c3 = rnd(1, 540, 960, 48).to(torch.bfloat16), rnd(1, 540, 960, 16).to(torch.bfloat16), rnd(1, 88, 158, 16).to(torch.bfloat16)
    wf, w4 = rnd(16, 16) * 0.3, rnd(48, 16) * 0.3
    for _ in range(REP):
        F_.esa_tail(x, c1, c3, wf, rnd(16) * 0.1, w4, rnd(48) * 0.1, 48)
else:
    n, h, w = 16, 1024, 1024
    x = rnd(n, h, w, 64).to(torch.bfloat16)
    w1, w2 = rnd(64, 64, 3, 3) * 0.1, rnd(64, 64, 3, 3) * 0.1
    wp1, wp2 = F_.conv3x3_pack_weights(w1, 64, 64), F_.conv3x3_pack_weights(w2, 64, 64)
    b1, b2 = rnd(64) * 0.1, rnd(64) * 0.1
    out, mid, bits = F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2, emit_bits=True)
    if case == "pairbits":
        for _ in range(REP - 1):
            F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2, emit_bits=True)
    else:
        wtb = F_.conv3x3_pack_weights(w2, 64, 64, transpose_flip=True)
        wta = F_.conv3x3_pack_weights(w1, 64, 64, transpose_flip=True)
        for _ in range(REP):
            F_.conv3x3_pair(x, wtb, None, wta, None, act2=F_.ACT_NONE, alpha=0.2, apply_bits=bits)
torch.cuda.synchronize()
