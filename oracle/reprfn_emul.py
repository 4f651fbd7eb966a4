"""CPU statement of RepRFN's training step as the HIP path computes it (core/arch/RepRFN.py:183-191,257-265,295-308,
327-339 under loss.backward(), core/trainer.py:160).  TEST INFRASTRUCTURE ONLY.

The graph is the one ``superresolution_amd.reprfn.RepRFN.forward_train`` assembles (every RepBlock as ONE 3x3
convolution whose kernel is merged from the nine branches by differentiable weight algebra; conv + LeakyReLU chains;
ESA with its strided convolution, max-pool, resize and gate), written with fp32 torch operators on the CPU.

Two modes:
  * ``emulate=False``: plain fp32.  This mode is pinned to the reference's own training-form run
    (tests/golden/reprfn_train.npz): same output and the same gradient for every parameter.
  * ``emulate=True``: every tensor the HIP path STORES as bf16 is rounded to bf16 at exactly that point: activations,
    activation gradients (one rounding per kernel output, so a mask fused into a data-gradient launch rounds once),
    and the merged / plain weights when they are packed for the MFMA kernel.  Accumulation stays fp32 and weight /
    bias gradients stay fp32, as on the GPU.  This is what the HIP gradients are held to at a tight tolerance: what is
    left between the two is fp32 summation order.

``forced``: a network that stores bf16 maps is not reproducible to the bit across summation orders -- a 1e-7 difference
flips a bf16 rounding now and then, each flip perturbs the 432 sums it feeds, and after a few layers a sizeable share
of all stored values differs by one bf16 step.  That is harmless in the forward pass, but the backward pass reads
discontinuous functions of those values (LeakyReLU slopes 1 / 0.05 around zero, max-pool positions): a handful of
flipped decisions moves a gradient tensor by several percent although nothing is wrong.  To check the BACKWARD code at
a tight tolerance the oracle can therefore be run on the stored maps of the GPU pass itself: ``forced[name]`` replaces
the oracle's own value of that map (names as in ``RepRFN.forward_train(trace=...)``), so both backward passes take the
same decisions and differ only by rounding noise.  The forward pass is checked separately (every map within a bf16
step or two of the unforced oracle).
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as TF
from torch import Tensor
from torch.autograd import Function

_SLOPE = 0.05


def _r(t: Tensor, emulate: bool) -> Tensor:
    return t.to(torch.bfloat16).float() if emulate else t


def _lrelu_mask(dy: Tensor, y: Tensor, slope: float) -> Tensor:
    return torch.where(y > 0, dy, dy * slope)


class _Conv(Function):
    """act(conv(x, W) + b): W rounded when packed, output rounded; backward: mask pass (rounded), fp32 weight / bias
    gradient from the rounded operands, data gradient rounded."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, lrelu, emulate, need_dx, force):
        wq = _r(w.detach(), emulate)
        y = TF.conv2d(x, wq, b.detach(), stride=stride, padding=padding)
        if lrelu:
            y = TF.leaky_relu(y, _SLOPE)
        y = _r(y, emulate) if force is None else force.clone()
        ctx.save_for_backward(x, wq, y)
        ctx.cfg = (stride, padding, lrelu, emulate, need_dx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq, y = ctx.saved_tensors
        stride, padding, lrelu, emulate, need_dx = ctx.cfg
        dz = _r(_lrelu_mask(dy, y, _SLOPE), emulate) if lrelu else dy
        dw = torch.nn.grad.conv2d_weight(x, wq.shape, dz, stride=stride, padding=padding)
        db = dz.sum(dim=(0, 2, 3))
        dx = None
        if need_dx:
            dx = _r(torch.nn.grad.conv2d_input(x.shape, wq, dz, stride=stride, padding=padding), emulate)
        return dx, dw, db, None, None, None, None, None, None


def _conv(x, w, b, emulate, stride=1, padding=1, lrelu=False, need_dx=True, force=None):
    return _Conv.apply(x, w, b, stride, padding, lrelu, emulate, need_dx, force)


class _Chain(Function):
    """y_i = lrelu(conv3x3(y_{i-1}, W_i) + b_i): the backward fuses the LeakyReLU backward of layer i-1 into the
    data-gradient launch of layer i, so that product is rounded ONCE."""

    @staticmethod
    def forward(ctx, x, emulate, forces, *wb):
        ws = [_r(w.detach(), emulate) for w in wb[0::2]]
        ys = [x]
        for i, (w, b) in enumerate(zip(ws, wb[1::2])):
            y = _r(TF.leaky_relu(TF.conv2d(ys[-1], w, b.detach(), padding=1), _SLOPE), emulate)
            ys.append(y if forces is None or forces[i] is None else forces[i].clone())
        ctx.save_for_backward(*ys, *ws)
        ctx.n, ctx.emulate = len(ws), emulate
        return ys[-1]

    @staticmethod
    def backward(ctx, dy):
        n, emulate = ctx.n, ctx.emulate
        ys, ws = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        grads = [None] * (2 * n)
        dz = _r(_lrelu_mask(dy, ys[n], _SLOPE), emulate)
        dx = None
        for i in range(n, 0, -1):
            w, x = ws[i - 1], ys[i - 1]
            grads[2 * (i - 1)] = torch.nn.grad.conv2d_weight(x, w.shape, dz, padding=1)
            grads[2 * (i - 1) + 1] = dz.sum(dim=(0, 2, 3))
            d = torch.nn.grad.conv2d_input(x.shape, w, dz, padding=1)
            if i > 1:
                dz = _r(_lrelu_mask(d, x, _SLOPE), emulate)
            else:
                dx = _r(d, emulate)
        return (dx, None, None) + tuple(grads)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b, emulate, force):
        return _r(a + b, emulate) if force is None else force.clone()

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, None, None


class _Gate(Function):
    @staticmethod
    def forward(ctx, x, gate, emulate, force):
        ctx.save_for_backward(x, gate)
        ctx.emulate = emulate
        return _r(x * torch.sigmoid(gate), emulate) if force is None else force.clone()

    @staticmethod
    def backward(ctx, dy):
        x, gate = ctx.saved_tensors
        s = torch.sigmoid(gate)
        return _r(dy * s, ctx.emulate), _r(dy * x * s * (1.0 - s), ctx.emulate), None, None


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k, s, emulate):
        y, idx = TF.max_pool2d(x, k, s, return_indices=True)
        ctx.save_for_backward(idx)
        ctx.cfg = (x.shape, k, s, emulate)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, k, s, emulate = ctx.cfg
        n, c, h, w = shape
        dx = torch.zeros(n, c, h * w, dtype=dy.dtype)
        dx.scatter_add_(2, idx.reshape(n, c, -1), dy.reshape(n, c, -1))     # overlapping windows: summed in fp32
        return _r(dx.view(shape), emulate), None, None, None


class _ResizeAdd(Function):
    @staticmethod
    def forward(ctx, small, addend, emulate, force):
        ctx.cfg = (small.shape, emulate)
        if force is not None:
            return force.clone()
        return _r(TF.interpolate(small, addend.shape[-2:], mode="bilinear", align_corners=False) + addend, emulate)

    @staticmethod
    def backward(ctx, dy):
        shape, emulate = ctx.cfg
        with torch.enable_grad():
            z = torch.zeros(shape, requires_grad=True)
            TF.interpolate(z, dy.shape[-2:], mode="bilinear", align_corners=False).backward(dy)
        return _r(z.grad, emulate), dy, None, None


class _Fan(Function):
    """Identity in front of a map with several consumers: the GPU pass adds their gradients as bf16 tensors."""

    @staticmethod
    def forward(ctx, x, emulate):
        ctx.emulate = emulate
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return _r(dy, ctx.emulate), None


class _ShuffleOut(Function):
    @staticmethod
    def forward(ctx, x, r, emulate):
        ctx.cfg = (r, emulate)
        return TF.pixel_shuffle(x, r)

    @staticmethod
    def backward(ctx, dy):
        r, emulate = ctx.cfg
        return _r(TF.pixel_unshuffle(dy, r), emulate), None, None


def forward(model, x: Tensor, emulate: bool, forced: Optional[Dict[str, Tensor]] = None, trace: Optional[dict] = None) -> Tensor:
    """``model``: a ``superresolution_amd.RepRFN`` in its training form on the CPU (it supplies the parameters and the
    differentiable merge of the nine branches); ``x`` (N,3,h,w) fp32.  Returns the (N,3,4h,4w) output with a graph.
    ``forced`` / ``trace``: stored maps (N,C,h,w) fp32 by stage name, taken from / written to (module docstring)."""
    from superresolution_amd.reprfn import merged_kernels_autograd
    e = emulate
    fz = (lambda name: None) if forced is None else (lambda name: forced.get(name))

    def rec(name, t):
        if trace is not None:
            trace[name] = t
        return t
    xin = _r(x.detach().float(), e)
    fea = _Fan.apply(rec("fea", _conv(xin, model.fea_conv.weight, model.fea_conv.bias, e, need_dx=False, force=fz("fea"))), e)
    out = fea
    merged = merged_kernels_autograd(getattr(getattr(model, f"reprfb{bi}"), f"repblock{ri}")
                                     for bi in range(1, 5) for ri in range(1, 4))
    for bi in range(1, 5):
        rfb = getattr(model, f"reprfb{bi}")
        p = f"b{bi}."
        layers = merged[(bi - 1) * 3:bi * 3] + [(rfb.conv3.weight, rfb.conv3.bias)]
        forces = None if forced is None else [forced.get(f"{p}chain{i + 1}") for i in range(len(layers))]
        ch = _Chain.apply(out, e, forces, *[q for wb in layers for q in wb])
        if trace is not None:
            for i, y_i in enumerate(ch.grad_fn.saved_tensors[1:len(layers) + 1]):
                trace[f"{p}chain{i + 1}"] = y_i
        t = _Fan.apply(rec(p + "t", _Add.apply(ch, out, e, fz(p + "t"))), e)
        esa = rfb.esa
        c1_ = _Fan.apply(rec(p + "c1_", _conv(t, esa.conv1.weight, esa.conv1.bias, e, padding=0, force=fz(p + "c1_"))), e)
        c1 = rec(p + "c1", _conv(c1_, esa.conv2.weight, esa.conv2.bias, e, stride=2, padding=0, force=fz(p + "c1")))
        v_max = rec(p + "vmax", _MaxPool.apply(c1, 7, 3, e))
        c3 = rec(p + "c3", _conv(v_max, esa.conv3.weight, esa.conv3.bias, e, force=fz(p + "c3")))
        cf = rec(p + "cf", _conv(c1_, esa.conv_f.weight, esa.conv_f.bias, e, padding=0, force=fz(p + "cf")))
        s = rec(p + "s", _ResizeAdd.apply(c3, cf, e, fz(p + "s")))
        c4 = rec(p + "c4", _conv(s, esa.conv4.weight, esa.conv4.bias, e, padding=0, force=fz(p + "c4")))
        out = _Fan.apply(rec(p + "out", _Gate.apply(t, c4, e, fz(p + "out"))), e)
    out = rec("lradd", _Add.apply(rec("lr", _conv(out, model.lr_conv.weight, model.lr_conv.bias, e, force=fz("lr"))), fea, e, fz("lradd")))
    c = rec("up", _conv(out, model.upsampler.conv.weight, model.upsampler.conv.bias, e, force=fz("up")))
    return _ShuffleOut.apply(c, model.upsampler.upscale_factor, e)


def forward_backward(model, x: Tensor, r: Tensor, emulate: bool, forced: Optional[Dict[str, Tensor]] = None,
                     trace: Optional[dict] = None) -> Tuple[Tensor, Dict[str, Tensor]]:
    """Output and the gradient of sum(y * r) for every trainable parameter of ``model`` (left untouched: gradients are
    returned, not accumulated)."""
    params = {n: p for n, p in model.named_parameters() if p.requires_grad}
    y = forward(model, x, emulate, forced, trace)
    grads = torch.autograd.grad((y * r).sum(), list(params.values()), allow_unused=True)
    return y.detach(), {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(params.items(), grads)}
