"""Differentiable operators over libsrhip.so for networks whose graph is assembled per operator (RepRFN):
each is a torch.autograd.Function whose forward AND backward run on the HIP kernels; torch only chains them
(and differentiates the small weight-space algebra of the re-parameterisation).

Activations are (N,H,W,C) bf16 tensors whose channel count is 16, 32 or 64 (the weight-gradient kernel's tile
sizes): a 48-channel feature map travels padded to 64 channels, the pad channels staying exactly zero through
every operator (zero weights, zero bias, LeakyReLU(0) = 0, x * sigmoid(.) = 0 for x = 0)."""
import os
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


class _Resize(Function):
    """F.interpolate(x, (ho, wo), mode="bilinear", align_corners=...) on an NHWC bf16 map; backward: the gather-form
    adjoint kernel (no atomics: repeats bit for bit)."""

    @staticmethod
    def forward(ctx, x: Tensor, ho: int, wo: int, align_corners: bool):
        ctx.hw, ctx.align = tuple(x.shape[1:3]), align_corners
        return F_.resize_bilinear_nhwc(x, ho, wo, align_corners=align_corners)

    @staticmethod
    def backward(ctx, dy: Tensor):
        return F_.resize_bilinear_bwd_nhwc(dy.contiguous(), ctx.hw[0], ctx.hw[1], align_corners=ctx.align), None, None, None


def resize_bilinear(x: Tensor, ho: int, wo: int, align_corners: bool = False) -> Tensor:
    return _Resize.apply(x, ho, wo, align_corners)


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


class _PixelShuffle4To16(Function):
    """(N,h,w,Cs) bf16 -> (N,4h,4w,16) bf16: nn.PixelShuffle(4) written straight into the next convolution's padded input
    layout (core/models/model_final.py:446-448), and its backward, one pass each."""

    @staticmethod
    def forward(ctx, x: Tensor, c: int):
        ctx.c, ctx.cs = c, x.shape[-1]
        return F_.pixel_shuffle4_nhwc_to_nhwc16(x, c)

    @staticmethod
    def backward(ctx, dy: Tensor):
        return F_.pixel_shuffle4_nhwc16_bwd(dy.contiguous(), ctx.c, ctx.cs), None


def pixel_shuffle4_to_nhwc16(x: Tensor, c: int) -> Tensor:
    return _PixelShuffle4To16.apply(x, c)


class _NhwcToNchwF32(Function):
    """The first c channels of an NHWC bf16 map as an fp32 NCHW tensor (the network's output), one pass each way."""

    @staticmethod
    def forward(ctx, x: Tensor, c: int):
        ctx.cs = x.shape[-1]
        return F_.nhwc_bf16_to_nchw_f32(x, c)

    @staticmethod
    def backward(ctx, dy: Tensor):
        return F_.nchw_f32_to_nhwc_bf16(dy.float().contiguous(), ctx.cs), None


def nhwc_to_nchw_f32(x: Tensor, c: int) -> Tensor:
    return _NhwcToNchwF32.apply(x, c)


# ------------------------------------------------------------------------------------------------
# Any layer functional.conv2d_auto runs (kernel 1 or 3, stride 1 or 2, any channel counts, up to three concatenated
# sources), differentiable: the temporal network's training graph (core/models/model_final.py under loss.backward()).
# ------------------------------------------------------------------------------------------------
def _tile_channels(c: int) -> int:
    """Channel count of a slice as the weight-gradient kernel tiles it (16, 32 or 64)."""
    return 16 if c <= 16 else (32 if c <= 32 else 64)


def _zero_insert2(dz: Tensor, h: int, w: int) -> Tensor:
    """(N,ho,wo,C) -> (N,h,w,C) with dz at the even positions: a stride-2 layer's output gradient seen from the
    stride-1 layer it is a 2x2 pick of (its data and weight gradients are that layer's on this tensor)."""
    if dz.dtype == torch.bfloat16 and dz.shape[-1] % 8 == 0 and dz.is_contiguous():
        return F_.zero_insert2_nhwc(dz, h, w)   # one pass instead of a fill and a strided copy
    up = torch.zeros((dz.shape[0], h, w, dz.shape[-1]), dtype=dz.dtype, device=dz.device)
    up[:, ::2, ::2] = dz
    return up


class _ConvAuto(Function):
    """act(nn.Conv2d(concat(srcs))) through functional.conv2d_auto.  Backward, per layer shape:
      * ReLU mask from the saved output;
      * data gradient: the same operator on transposed (3x3: and flipped) weights -- for stride 2 applied to the output
        gradient with zeros inserted at the odd positions -- then cut into the sources' channel ranges;
      * weight and bias gradient: the MFMA weight-gradient kernel, one launch per (<= 64 input, <= 64 output) channel
        slice pair, stride 2 again through the zero-inserted gradient, a 1x1 layer as the centre tap of a 3x3 one."""

    @staticmethod
    def forward(ctx, weight: Tensor, bias: Tensor, stride: int, padding: int, act: int, *srcs: Tensor):
        k = weight.shape[-1]
        if bias is None:
            raise ValueError("autograd_ops.conv2d: a bias tensor is required (every layer of the path has one)")
        if weight.shape[-2] != k or k not in (1, 3) or padding != k // 2 or stride not in (1, 2):
            raise ValueError(f"autograd_ops.conv2d: differentiable layers are kernel 1 or 3 with padding kernel // 2 and "
                             f"stride 1 or 2 (the backward derives its data gradient from that); got kernel "
                             f"{tuple(weight.shape[-2:])}, stride {stride}, padding {padding}")
        w = weight.detach().float().contiguous()
        y = F_.conv2d_auto(tuple(srcs) if len(srcs) > 1 else srcs[0], w, bias.detach().float().contiguous(), stride=stride,
                           padding=padding, act=act)
        ctx.save_for_backward(w, y if act != F_.ACT_NONE else None, *srcs)
        ctx.cfg = (stride, padding, act)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        w, y = ctx.saved_tensors[:2]
        srcs = ctx.saved_tensors[2:]
        stride, padding, act = ctx.cfg
        n, h, wd = srcs[0].shape[:3]
        dz = dy.contiguous() if y is None else F_.relu_mask(dy.contiguous(), y)
        dzs = dz if stride == 1 else _zero_insert2(dz, h, wd)         # gradient of the equivalent stride-1 layer
        dw, db = _auto_wgrad(srcs, dzs, w)
        dsrcs = [None] * len(srcs)
        if any(ctx.needs_input_grad[5:]):
            dsrcs = _auto_dgrad_sources(dzs, w, srcs, ctx.needs_input_grad[5:])
        return (dw, db, None, None, None) + tuple(dsrcs)


def _auto_wgrad(srcs, dzs: Tensor, w: Tensor):
    """Weight and bias gradient of a layer of ``_ConvAuto`` from the gradient of its (equivalent stride-1) pre-activation:
    the MFMA weight-gradient kernel per (<= 64 input, <= 64 output) channel slice pair; a 1x1 layer is the centre tap of
    the 3x3 result (a GEMM over a million pixels with a 160 x 160 output is a shape the GEMM library serves three times
    slower than that).  The kernel reads the channel windows in place: no slice of x or dz is copied out."""
    cout, cin, k, _ = w.shape
    # (every entry is written: the slice pairs tile the whole weight, the first input slice's jobs the whole bias)
    # (a 1x1 layer: the kernel's centre-tap form writes the (O,I,1,1) gradient directly; SRHIP_NO_WGRAD_1X1=1: as the centre
    # tap of the 3x3 result, eight taps computed and dropped)
    k_out = 1 if (k == 1 and os.environ.get("SRHIP_NO_WGRAD_1X1", "0") != "1") else 3
    dw3 = torch.empty((cout, cin, k_out, k_out), dtype=torch.float32, device=w.device)
    db = torch.empty((cout,), dtype=torch.float32, device=w.device)
    dz_slices = [(o, min(64, cout - o)) for o in range(0, cout, 64)]
    # all slice pairs of one tile shape go out together (one gradient launch + one reduction per group of <= 32 pairs
    # instead of one each per pair: on the 1/4 and 1/8 resolution maps a pair is a few microseconds of matrix work)
    groups = {}
    k0 = 0
    for t in srcs:
        for a in range(0, t.shape[-1], 64):
            ci = min(a + 64, t.shape[-1]) - a
            ci_real = min(ci, cin - (k0 + a))
            if ci_real <= 0:
                continue
            for o, co in dz_slices:
                # the bias gradient rides on the first input slice's job (dz times ones on the MFMA pipe)
                groups.setdefault((_tile_channels(ci), _tile_channels(co)), []).append(
                    (t, a, ci_real, k0 + a, dzs, o, co, o, k0 + a == 0))
        k0 += t.shape[-1]
    batch = 1 if os.environ.get("SRHIP_NO_WGRAD_MULTI", "0") == "1" else F_.WGRAD_MAX_JOBS
    for (ci_tile, co_tile), jobs in groups.items():
        for b in range(0, len(jobs), batch):
            F_.conv3x3_wgrad_multi(jobs[b:b + batch], dw3, db, ci_tile, co_tile)
    return (dw3 if k == k_out else dw3[:, :, 1:2, 1:2].contiguous()), db


def _dgrad_rows(dzs: Tensor, w: Tensor, row0: int, rows: int, mask: Tensor = None) -> Tensor:
    """Data gradient w.r.t. input channels [row0, row0 + rows) of a layer: the same operator on transposed (3x3: flipped)
    weights; ``mask`` (the ReLU output of the layer in front) is applied by the launch's epilogue.  On the implicit-GEMM
    kernel the transposed weights are packed straight from the stored ones; a layer that is one launch of the
    register-resident kernel (<= 64 channels both ways) materialises them for that kernel's packer."""
    k, cpad = w.shape[-1], dzs.shape[-1]
    if F_.use_conv_gemm and (cpad > 64 or rows > 64) and cpad % 16 == 0 and os.environ.get("SRHIP_NO_PACK_T", "0") != "1":
        wp = F_.conv_gemm_pack_weights_transposed(w, row0, rows, cpad)
        return F_.conv_gemm(dzs, wp, None, rows, k, stride=1, padding=k // 2, res=mask,
                            res_mode=F_.RES_MASK if mask is not None else F_.RES_ADD)
    if k == 3 and cpad <= 64 and rows <= 64 and cpad % 16 == 0 and w.dtype == torch.float32 and w.is_contiguous() \
            and cpad - w.shape[0] < 16 and (mask is None or mask.shape[-1] == F_.pad16(rows)) \
            and os.environ.get("SRHIP_NO_PACK_T", "0") != "1":
        # one launch of the register-resident kernel: its packer reads the stored weights transposed and flipped (no flip /
        # copy launches in front of it)
        wp = F_.conv3x3_pack_weights(w, w.shape[0], cpad, cout=rows, cout_off=row0, transpose_flip=True)
        return F_.conv3x3(dzs, wp, None, rows, res=mask, res_mode=F_.RES_MASK if mask is not None else F_.RES_ADD)
    wt = (w.transpose(0, 1).flip(2, 3) if k == 3 else w.transpose(0, 1))[row0:row0 + rows].contiguous()
    if mask is None:
        return F_.conv2d_auto(dzs, wt, None, stride=1, padding=k // 2)        # (N,h,w,pad16(rows))
    return F_.conv2d_auto(dzs, wt, None, stride=1, padding=k // 2, res=mask, res_mode=F_.RES_MASK)


def _auto_dgrad(dzs: Tensor, w: Tensor, mask: Tensor = None) -> Tensor:
    """Data gradient w.r.t. the concatenated input."""
    return _dgrad_rows(dzs, w, 0, w.shape[1], mask)


def _auto_dgrad_sources(dzs: Tensor, w: Tensor, srcs, needs):
    """Data gradients of a layer's sources.  A concatenated input gets one launch per source on that source's rows of the
    transposed weights, each writing the source's gradient map directly (same values as one launch over all input
    channels cut apart afterwards, without the zero fill and the strided copy per source)."""
    cin, k = w.shape[1], w.shape[-1]
    widths = [t.shape[-1] for t in srcs]
    real = [max(0, min(c, cin - sum(widths[:i]))) for i, c in enumerate(widths)]
    if len(srcs) == 1 or any(F_.pad16(r) != c for r, c in zip(real, widths)):
        return _split_sources(_auto_dgrad(dzs, w), srcs, cin, needs)
    out, k0 = [None] * len(srcs), 0
    for i, r in enumerate(real):
        if needs[i]:
            out[i] = _dgrad_rows(dzs, w, k0, r)
        k0 += widths[i]
    return out


def _split_sources(dx: Tensor, srcs, cin: int, needs):
    """The gradient of a concatenated input cut into the sources' channel ranges."""
    out, k0 = [None] * len(srcs), 0
    for i, t in enumerate(srcs):
        c = t.shape[-1]
        if needs[i]:
            if k0 == 0 and dx.shape[-1] == c:
                out[i] = dx
            else:
                d = torch.zeros_like(t)
                hi = min(k0 + c, dx.shape[-1])
                d[..., :hi - k0] = dx[..., k0:hi]
                out[i] = d
        k0 += c
    return out


class _ConvReluChain(Function):
    """y_i = relu(conv_i(y_{i-1}) + b_i), i = 1..n, as ONE autograd node; y_0 may be several maps read as their
    concatenation.  The intermediate maps have no other consumer, so each intermediate ReLU backward rides on the
    data-gradient launch of the layer behind it (mask epilogue) instead of a pass of its own: one mask pass per chain.
    Same launches, same roundings as n ``_ConvAuto`` nodes (a mask only zeroes): gradients are bit-equal."""

    @staticmethod
    def forward(ctx, nsrc: int, cfgs, *tensors: Tensor):
        srcs, wb = tensors[:nsrc], tensors[nsrc:]
        ws = [w.detach().float().contiguous() for w in wb[0::2]]
        ys, x = [], (tuple(srcs) if nsrc > 1 else srcs[0])
        for w, b, (stride, padding) in zip(ws, wb[1::2], cfgs):
            k = w.shape[-1]
            if b is None or w.shape[-2] != k or k not in (1, 3) or padding != k // 2 or stride not in (1, 2):
                raise ValueError("autograd_ops.conv_relu_chain: layers are kernel 1 or 3 with padding kernel // 2, stride 1 or 2, with bias")
            x = F_.conv2d_auto(x, w, b.detach().float().contiguous(), stride=stride, padding=padding, act=F_.ACT_RELU)
            ys.append(x)
        ctx.save_for_backward(*srcs, *ys, *ws)
        ctx.nsrc, ctx.cfgs = nsrc, tuple(cfgs)
        return ys[-1]

    @staticmethod
    def backward(ctx, dy: Tensor):
        nsrc, cfgs = ctx.nsrc, ctx.cfgs
        n = len(cfgs)
        saved = ctx.saved_tensors
        srcs, ys, ws = saved[:nsrc], saved[nsrc:nsrc + n], saved[nsrc + n:]
        grads = [None] * (2 * n)
        dz = F_.relu_mask(dy.contiguous(), ys[n - 1])
        dsrcs = [None] * nsrc
        for i in range(n - 1, -1, -1):
            w = ws[i]
            inp = srcs if i == 0 else (ys[i - 1],)
            h, wd = inp[0].shape[1:3]
            dzs = dz if cfgs[i][0] == 1 else _zero_insert2(dz, h, wd)
            grads[2 * i], grads[2 * i + 1] = _auto_wgrad(inp, dzs, w)
            if i > 0:
                dz = _auto_dgrad(dzs, w, mask=ys[i - 1])
                if dz.shape[-1] != ys[i - 1].shape[-1]:
                    dz = _split_sources(dz, (ys[i - 1],), w.shape[1], (True,))[0]
            elif any(ctx.needs_input_grad[2:2 + nsrc]):
                dsrcs = _auto_dgrad_sources(dzs, w, srcs, ctx.needs_input_grad[2:2 + nsrc])
        return (None, None) + tuple(dsrcs) + tuple(grads)


def conv_relu_chain(x, layers) -> Tensor:
    """layers: [(weight, bias, stride, padding), ...], each followed by ReLU; x an NHWC bf16 map or a tuple of maps."""
    srcs = tuple(x) if isinstance(x, (tuple, list)) else (x,)
    flat = [t for (w, b, _, _) in layers for t in (w, b)]
    return _ConvReluChain.apply(len(srcs), tuple((s, p) for (_, _, s, p) in layers), *srcs, *flat)


class _AttentionFuse(Function):
    """cur + sigmoid(f_self * g_cross) * fmap + sigmoid(g_self * f_cross) * grad (core/models/model_final.py:422) as one
    launch forward and one backward; the gates are channel 0 of padded (N,H,W,Cg) maps and receive the gradient of the
    whole map (zeros beside channel 0)."""

    @staticmethod
    def forward(ctx, cur, fmap, grad, f_self, g_cross, g_self, f_cross):
        ctx.save_for_backward(fmap, grad, f_self, g_cross, g_self, f_cross)
        return F_.attention_fuse(cur, fmap, grad, f_self[..., 0:1], g_cross[..., 0:1], g_self[..., 0:1], f_cross[..., 0:1])

    @staticmethod
    def backward(ctx, d: Tensor):
        fmap, grad, f_self, g_cross, g_self, f_cross = ctx.saved_tensors
        d = d.contiguous()
        out = F_.attention_fuse_bwd(d, fmap, grad, f_self[..., 0:1], g_cross[..., 0:1], g_self[..., 0:1], f_cross[..., 0:1],
                                    gate_channels=f_self.shape[-1])
        return (d,) + out


def attention_fuse(cur: Tensor, fmap: Tensor, grad: Tensor, f_self: Tensor, g_cross: Tensor, g_self: Tensor, f_cross: Tensor) -> Tensor:
    """Differentiable functional.attention_fuse; the four gates are padded maps of one width whose channel 0 counts."""
    if len({t.shape[-1] for t in (f_self, g_cross, g_self, f_cross)}) != 1:
        raise ValueError("attention_fuse: the gate maps must have one channel count")
    return _AttentionFuse.apply(cur, fmap, grad, f_self, g_cross, g_self, f_cross)


class _ScaleAdd(Function):
    """bf16(a + scale * y) with a trained 0-d fp32 scale (fmap + enhance_param * enhance, core/models/model_final.py:256-258):
    one pass forward; backward one pass for dy = bf16(scale * d) and dscale = sum(d * y) in fp32, da = d."""

    @staticmethod
    def forward(ctx, a, y, scale):
        ctx.save_for_backward(y, scale)
        return F_.scale_add_bf16(a, y, scale.detach())

    @staticmethod
    def backward(ctx, d: Tensor):
        y, scale = ctx.saved_tensors
        d = d.contiguous()
        dy, ds = F_.scale_add_bwd_bf16(d, y, scale.detach())
        return d, dy, ds.reshape(scale.shape).to(scale.dtype)


def scale_add(a: Tensor, y: Tensor, scale: Tensor) -> Tensor:
    """Differentiable functional.scale_add_bf16: gradients for both maps and for the scale."""
    return _ScaleAdd.apply(a, y, scale)


def conv2d(x, weight: Tensor, bias: Tensor, stride: int = 1, padding: int = 1, act: int = F_.ACT_NONE) -> Tensor:
    """Differentiable functional.conv2d_auto; x is an NHWC bf16 map or a tuple of maps read as their concatenation."""
    srcs = tuple(x) if isinstance(x, (tuple, list)) else (x,)
    return _ConvAuto.apply(weight, bias, stride, padding, act, *srcs)
