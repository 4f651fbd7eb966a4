"""Differentiable operators over libsrhip.so for networks whose graph is assembled per operator (RepRFN):
each is a torch.autograd.Function whose forward AND backward run on the HIP kernels; torch only chains them
(and differentiates the small weight-space algebra of the re-parameterisation).

Activations are (N,H,W,C) bf16 tensors whose channel count is 16, 32 or 64 (the weight-gradient kernel's tile
sizes): a 48-channel feature map travels padded to 64 channels, the pad channels staying exactly zero through
every operator (zero weights, zero bias, LeakyReLU(0) = 0, x * sigmoid(.) = 0 for x = 0)."""
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import functional as F_


class _Conv3x3(Function):
    """act(conv3x3(x, W) + b) with W an fp32 (O,I,3,3) tensor that may itself be the result of differentiable
    weight algebra; backward: LeakyReLU mask from the saved output, weight / bias gradient (MFMA, deterministic
    reduction), data gradient as the same convolution on transposed, flipped weights."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, act: int, slope: float, cout_stride: Optional[int]):
        w = weight.detach().float().contiguous()
        cout, cin = w.shape[0], w.shape[1]
        y = F_.conv3x3(x, F_.conv3x3_pack_weights(w, cin, x.shape[-1]), bias.detach().float().contiguous(), cout, act=act,
                       slope=slope, cout_stride=cout_stride)
        ctx.save_for_backward(x, w, y if act != F_.ACT_NONE else None)
        ctx.act, ctx.slope = act, (slope if act == F_.ACT_LRELU else 0.0)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w, y = ctx.saved_tensors
        cout, cin = w.shape[0], w.shape[1]
        dz = dy.contiguous() if y is None else F_.lrelu_mask(dy, y, ctx.slope)
        dw, db = torch.empty_like(w), torch.empty((cout,), dtype=torch.float32, device=w.device)
        F_.conv3x3_wgrad(x, dz, dw, db, cin, cout)
        dx = None
        if ctx.needs_input_grad[0]:
            wp = F_.conv3x3_pack_weights(w, cout, dz.shape[-1], cout=cin, transpose_flip=True)
            dx = F_.conv3x3(dz, wp, None, cin, cout_stride=x.shape[-1])
        return dx, dw, db, None, None, None


def conv3x3(x: Tensor, weight: Tensor, bias: Tensor, act: int = F_.ACT_NONE, slope: float = 0.0,
            cout_stride: Optional[int] = None) -> Tensor:
    return _Conv3x3.apply(x, weight, bias, act, slope, cout_stride)


class _ConvLReluChain(Function):
    """y_i = lrelu(conv3x3(y_{i-1}, W_i) + b_i), i = 1..n, as ONE autograd node (inputs: x, W_1, b_1, ..., W_n, b_n).
    The intermediate maps have no other consumer, so the backward fuses each intermediate LeakyReLU backward into the
    data-gradient launch of the layer behind it (MASK epilogue with the slope): one mask pass for the chain's output
    instead of one per layer."""

    @staticmethod
    def forward(ctx, x: Tensor, slope: float, cout_stride: Optional[int], *wb: Tensor):
        ws = [w.detach().float().contiguous() for w in wb[0::2]]
        ys = [x]
        for w, b in zip(ws, wb[1::2]):
            cout, cin = w.shape[0], w.shape[1]
            ys.append(F_.conv3x3(ys[-1], F_.conv3x3_pack_weights(w, cin, ys[-1].shape[-1]), b.detach().float().contiguous(), cout,
                                 act=F_.ACT_LRELU, slope=slope, cout_stride=cout_stride))
        ctx.save_for_backward(*ys, *ws)
        ctx.n, ctx.slope = len(ws), slope
        return ys[-1]

    @staticmethod
    def backward(ctx, dy: Tensor):
        n = ctx.n
        ys, ws = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        grads = [None] * (2 * n)
        dz = F_.lrelu_mask(dy.contiguous(), ys[n], ctx.slope)            # d(pre-activation) of the last layer
        dx = None
        for i in range(n, 0, -1):
            w, x = ws[i - 1], ys[i - 1]
            cout, cin = w.shape[0], w.shape[1]
            dw, db = torch.empty_like(w), torch.empty((cout,), dtype=torch.float32, device=w.device)
            F_.conv3x3_wgrad(x, dz, dw, db, cin, cout)
            grads[2 * (i - 1)], grads[2 * (i - 1) + 1] = dw, db
            if i > 1 or ctx.needs_input_grad[0]:
                wp = F_.conv3x3_pack_weights(w, cout, dz.shape[-1], cout=cin, transpose_flip=True)
                if i > 1:   # x is the LeakyReLU output of layer i-1: its backward rides on this launch
                    dz = F_.conv3x3(dz, wp, None, cin, cout_stride=x.shape[-1], res=x, res_mode=F_.RES_MASK, slope=ctx.slope)
                else:
                    dx = F_.conv3x3(dz, wp, None, cin, cout_stride=x.shape[-1])
        return (dx, None, None) + tuple(grads)


def conv3x3_lrelu_chain(x: Tensor, layers, slope: float, cout_stride: Optional[int] = None) -> Tensor:
    """layers: [(weight, bias), ...] applied in order, each followed by LeakyReLU(slope)."""
    flat = [t for wb in layers for t in wb]
    return _ConvLReluChain.apply(x, slope, cout_stride, *flat)


def conv1x1(x: Tensor, weight: Tensor, bias: Tensor, cout_stride: Optional[int] = None) -> Tensor:
    """1x1 convolution as the centre tap of a 3x3 kernel of zeros; autograd slices the centre tap of the 3x3
    weight gradient back out."""
    return conv3x3(x, torch.nn.functional.pad(weight, (1, 1, 1, 1)), bias, cout_stride=cout_stride)


class _Add(Function):
    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        return F_.add_bf16(a, b)

    @staticmethod
    def backward(ctx, dy: Tensor):
        return dy, dy


def add(a: Tensor, b: Tensor) -> Tensor:
    return _Add.apply(a, b)


class _SigmoidGate(Function):
    @staticmethod
    def forward(ctx, x: Tensor, gate: Tensor):
        ctx.save_for_backward(x, gate)
        return F_.sigmoid_gate(x, gate)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, gate = ctx.saved_tensors
        return F_.sigmoid_gate_bwd(x, gate, dy.contiguous())


def sigmoid_gate(x: Tensor, gate: Tensor) -> Tensor:
    return _SigmoidGate.apply(x, gate)


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x: Tensor, k: int, s: int):
        y, arg = F_.max_pool2d_arg_nhwc(x, k, s)
        ctx.save_for_backward(arg)
        ctx.k, ctx.s, ctx.hw = k, s, tuple(x.shape[1:3])
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (arg,) = ctx.saved_tensors
        return F_.max_pool2d_bwd_arg_nhwc(arg, dy.contiguous(), ctx.hw[0], ctx.hw[1], ctx.k, ctx.s), None, None


def max_pool2d(x: Tensor, k: int, s: int) -> Tensor:
    return _MaxPool.apply(x, k, s)


class _ResizeAdd(Function):
    """bilinear resize of `small` to the size of `addend` (align_corners=False) + addend."""

    @staticmethod
    def forward(ctx, small: Tensor, addend: Tensor):
        ctx.hw = tuple(small.shape[1:3])
        return F_.resize_bilinear_nhwc(small, addend.shape[1], addend.shape[2], addend=addend)

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = dy.contiguous()
        return F_.resize_bilinear_bwd_nhwc(dy, ctx.hw[0], ctx.hw[1]), dy


def resize_add(small: Tensor, addend: Tensor) -> Tensor:
    return _ResizeAdd.apply(small, addend)


class _PixelShuffleOut(Function):
    """(N,h,w,Cs) bf16 -> (N,c,h*r,w*r) fp32 (nn.PixelShuffle on the network's last feature map)."""

    @staticmethod
    def forward(ctx, x: Tensor, c: int, r: int):
        ctx.c, ctx.r, ctx.cs = c, r, x.shape[-1]
        return F_.pixel_shuffle_nhwc_to_nchw_f32(x, c, r)

    @staticmethod
    def backward(ctx, dy: Tensor):
        d = F_.pixel_unshuffle(dy.float().contiguous(), ctx.r)           # (N, c*r*r, h, w) fp32
        return F_.nchw_f32_to_nhwc_bf16(d, ctx.cs), None, None


def pixel_shuffle_out(x: Tensor, c: int, r: int) -> Tensor:
    return _PixelShuffleOut.apply(x, c, r)
