#!/usr/bin/env python3
"""Stage-by-stage comparison of RepRFN's training step on the HIP path with the bf16-storage oracle
(oracle/reprfn_emul.py): forward activations and their gradients at every operator boundary."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import reprfn_emul as E
from superresolution_amd import RepRFN, functional as F_, autograd_ops as A
from superresolution_amd.reprfn import merged_kernels_autograd, _SLOPE

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reprfn_train.npz"))


def model():
    torch.manual_seed(21)
    m = RepRFN(3, 3, 48, 4, deploy=False)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.requires_grad:
                p.copy_(p.to(torch.bfloat16).float())
    return m


def hip_forward(self, inputs, tr):
    def T(name, t):
        t.retain_grad()
        tr.append((name, t))
        return t
    cs = 64
    x = F_.nchw_f32_to_nhwc_bf16(inputs.detach().float())
    fea = T("fea", A.conv3x3(x, self.fea_conv.weight, self.fea_conv.bias, cout_stride=cs))
    out = fea
    merged = merged_kernels_autograd(getattr(getattr(self, f"reprfb{bi}"), f"repblock{ri}") for bi in range(1, 5) for ri in range(1, 4))
    for bi in range(1, 5):
        rfb = getattr(self, f"reprfb{bi}")
        layers = merged[(bi - 1) * 3:bi * 3] + [(rfb.conv3.weight, rfb.conv3.bias)]
        ch = T(f"b{bi}.chain", A.conv3x3_lrelu_chain(out, layers, _SLOPE, cout_stride=cs))
        t = T(f"b{bi}.t", A.add(ch, out))
        esa = rfb.esa
        c1_ = T(f"b{bi}.c1_", A.conv1x1(t, esa.conv1.weight, esa.conv1.bias))
        full = A.conv3x3(c1_, esa.conv2.weight, esa.conv2.bias)
        ho, wo = (full.shape[1] - 3) // 2 + 1, (full.shape[2] - 3) // 2 + 1
        c1 = T(f"b{bi}.c1", full[:, 1:2 * ho:2, 1:2 * wo:2, :].contiguous())
        vmax = T(f"b{bi}.vmax", A.max_pool2d(c1, 7, 3))
        c3 = T(f"b{bi}.c3", A.conv3x3(vmax, esa.conv3.weight, esa.conv3.bias))
        cf = T(f"b{bi}.cf", A.conv1x1(c1_, esa.conv_f.weight, esa.conv_f.bias))
        s = T(f"b{bi}.s", A.resize_add(c3, cf))
        c4 = T(f"b{bi}.c4", A.conv1x1(s, esa.conv4.weight, esa.conv4.bias, cout_stride=cs))
        out = T(f"b{bi}.out", A.sigmoid_gate(t, c4))
    lrc = T("lr", A.conv3x3(out, self.lr_conv.weight, self.lr_conv.bias, cout_stride=cs))
    out = T("lradd", A.add(lrc, fea))
    up = self.upsampler
    c = T("up", A.conv3x3(out, up.conv.weight, up.conv.bias, cout_stride=64))
    return A.pixel_shuffle_out(c, self.out_channels, up.upscale_factor)


def emu_forward(model, x, e, tr):
    def T(name, t):
        t.retain_grad()
        tr.append((name, t))
        return t
    xin = E._r(x.detach().float(), e)
    fea = T("fea", E._conv(xin, model.fea_conv.weight, model.fea_conv.bias, e, need_dx=False))
    out = fea
    merged = merged_kernels_autograd(getattr(getattr(model, f"reprfb{bi}"), f"repblock{ri}") for bi in range(1, 5) for ri in range(1, 4))
    for bi in range(1, 5):
        rfb = getattr(model, f"reprfb{bi}")
        layers = merged[(bi - 1) * 3:bi * 3] + [(rfb.conv3.weight, rfb.conv3.bias)]
        ch = T(f"b{bi}.chain", E._Chain.apply(out, e, *[p for wb in layers for p in wb]))
        t = T(f"b{bi}.t", E._Add.apply(ch, out, e))
        esa = rfb.esa
        c1_ = T(f"b{bi}.c1_", E._conv(t, esa.conv1.weight, esa.conv1.bias, e, padding=0))
        c1 = T(f"b{bi}.c1", E._conv(c1_, esa.conv2.weight, esa.conv2.bias, e, stride=2, padding=0))
        vmax = T(f"b{bi}.vmax", E._MaxPool.apply(c1, 7, 3, e))
        c3 = T(f"b{bi}.c3", E._conv(vmax, esa.conv3.weight, esa.conv3.bias, e))
        cf = T(f"b{bi}.cf", E._conv(c1_, esa.conv_f.weight, esa.conv_f.bias, e, padding=0))
        s = T(f"b{bi}.s", E._ResizeAdd.apply(c3, cf, e))
        c4 = T(f"b{bi}.c4", E._conv(s, esa.conv4.weight, esa.conv4.bias, e, padding=0))
        out = T(f"b{bi}.out", E._Gate.apply(t, c4, e))
    lrc = T("lr", E._conv(out, model.lr_conv.weight, model.lr_conv.bias, e))
    out = T("lradd", E._Add.apply(lrc, fea, e))
    c = T("up", E._conv(out, model.upsampler.conv.weight, model.upsampler.conv.bias, e))
    return E._ShuffleOut.apply(c, model.upsampler.upscale_factor, e)


def main():
    x, r = torch.from_numpy(z["x"]), torch.from_numpy(z["r"])
    mc, mg = model(), model().cuda().train()
    tre, trh = [], []
    ye = emu_forward(mc, x, True, tre)
    (ye * r).sum().backward()
    yh = hip_forward(mg, x.cuda(), trh)
    (yh * r.cuda()).sum().backward()
    print(f"{'stage':12s} {'fwd max|d|':>11s} {'fwd n!=':>8s} {'grad relL2':>11s} {'grad n!=':>9s}")
    for (n1, a), (n2, b) in zip(tre, trh):
        assert n1 == n2
        bb = b.detach().float().cpu().permute(0, 3, 1, 2)[:, :a.shape[1]]
        gb = b.grad.float().cpu().permute(0, 3, 1, 2)[:, :a.shape[1]]
        ga = a.grad
        fd = (a.detach() - bb).abs()
        gd = (ga - gb)
        print(f"{n1:12s} {float(fd.max()):11.3e} {int((fd > 0).sum()):8d} {float(gd.norm() / (ga.norm() + 1e-30)):11.3e} {int((gd != 0).sum()):9d} / {gd.numel()}")


if __name__ == "__main__":
    main()
