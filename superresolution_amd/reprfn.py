"""Drop-in for ``core.arch.RepRFN.RepRFN`` (core/arch/RepRFN.py:311-339) on libsrhip.so.

Contract kept: ctor ``RepRFN(in_channels=3, out_channels=3, feature_nums=48, sr_scale=4,
deploy=False)`` (YAML ``configs/trainer/4x_RepRFN.yaml:8-13``), ``forward(Tensor) -> Tensor``
(B,3,h,w) -> (B,3,4h,4w) fp32, identical ``state_dict`` keys and shapes in both the training form
(nine branches per RepBlock, RepRFN.py:163-181) and the deploy form (one ``rbr_reparam`` 3x3,
RepRFN.py:158-161), the same construction order (hence the same seed-for-seed initialisation), and
``switch_to_deploy()`` on every RepBlock (RepRFN.py:193-208).

Execution: every RepBlock runs as ONE 3x3 MFMA convolution + LeakyReLU(0.05).  In the training
form the equivalent 3x3 kernel is re-parameterised from the nine branches each time the weights
change (a 48x48x3x3 tensor: weight algebra, not activation work), which gives the deploy result for
a training-form state dict without materialising nine branch outputs.  The ESA block
(RepRFN.py:281-308) runs on the small-convolution, pooling, resize and gating kernels.
With autograd enabled the module runs `forward_train`: the same graph from differentiable operators
(`autograd_ops.py`), each RepBlock still ONE convolution whose merged kernel is differentiated back to the nine
branches.
"""
import os
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import functional as F_

_SLOPE = 0.05  # activation('lrelu') default, RepRFN.py:17,22


def _conv_layer(cin: int, cout: int, k: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, k, stride, (k - 1) // 2)


class SeqConv3x3(nn.Module):
    """1x1 convolution followed by a fixed per-channel 3x3 stencil with learnable scale
    (RepRFN.py:36-146); here a parameter container with the re-parameterisation rule."""
    _STENCILS = {
        "conv1x1-sobelx": [[1.0, 0.0, -1.0], [2.0, 0.0, -2.0], [1.0, 0.0, -1.0]],
        "conv1x1-sobely": [[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]],
        "conv1x1-laplacian": [[0.0, 1.0, 0.0], [1.0, -4.0, 1.0], [0.0, 1.0, 0.0]],
    }

    def __init__(self, seq_type: str, inp_planes: int, out_planes: int) -> None:
        super().__init__()
        if seq_type not in self._STENCILS:
            raise ValueError("the type of seqconv is not supported!")
        self.type, self.inp_planes, self.out_planes = seq_type, inp_planes, out_planes
        conv0 = nn.Conv2d(inp_planes, out_planes, kernel_size=1, padding=0)
        self.k0 = conv0.weight
        self.b0 = conv0.bias
        self.scale = nn.Parameter(torch.randn(size=(out_planes, 1, 1, 1)) * 1e-3)
        self.bias = nn.Parameter(torch.reshape(torch.randn(out_planes) * 1e-3, (out_planes,)))
        stencil = torch.tensor(self._STENCILS[seq_type], dtype=torch.float32)
        self.mask = nn.Parameter(stencil.expand(out_planes, 1, 3, 3).clone(), requires_grad=False)

    def rep_params(self) -> Tuple[Tensor, Tensor]:
        # the depthwise stencil after a 1x1: K[o,i,:,:] = scale[o] * mask[o] * k0[o,i]; the 1x1's bias
        # sees the stencil everywhere (the reference pads the 1x1 output with that bias): b0[o] * sum(stencil)
        dw = (self.scale * self.mask)[:, 0]                                  # (O,3,3)
        k = dw[:, None, :, :] * self.k0[:, :, 0, 0][:, :, None, None]       # (O,I,3,3)
        b = self.b0 * dw.sum(dim=(1, 2)) + self.bias
        return k, b


_IDENT = {}


def _identity_kernel(out_channels: int, in_channels: int, device, dtype) -> Tensor:
    """The identity branch as a 3x3 kernel (RepRFN.py:236-241): 1 at the centre tap of channel i -> i.  Built once per
    shape and device (constant; building it per step would put an index write into every training step)."""
    key = (out_channels, in_channels, str(device), dtype)
    hit = _IDENT.get(key)
    if hit is None:
        hit = torch.zeros(out_channels, in_channels, 3, 3, device=device, dtype=dtype)
        idx = torch.arange(min(in_channels, out_channels), device=device)
        hit[idx, idx, 1, 1] = 1.0
        _IDENT[key] = hit
    return hit


class RepBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, act_type: str = "lrelu", deploy: bool = False) -> None:
        super().__init__()
        self.deploy, self.in_channels, self.out_channels = deploy, in_channels, out_channels
        c, o = in_channels, out_channels
        if deploy:
            self.rbr_reparam = nn.Conv2d(c, o, (3, 3), 1, 1)
        else:
            self.rbr_3x3_branch = nn.Conv2d(c, o, (3, 3), 1, 1)
            self.rbr_3x1_branch = nn.Conv2d(c, o, (3, 1), 1, (1, 0))
            self.rbr_1x3_branch = nn.Conv2d(c, o, (1, 3), 1, (0, 1))
            self.rbr_1x1_branch = nn.Conv2d(c, o, (1, 1), 1, (0, 0))
            self.rbr_1x1_3x3_branch_1x1 = nn.Conv2d(c, 2 * c, (1, 1), 1, (0, 0), bias=False)
            self.rbr_1x1_3x3_branch_3x3 = nn.Conv2d(2 * c, o, (3, 3), 1, 1, bias=False)
            self.rbr_conv1x1_sbx_branch = SeqConv3x3("conv1x1-sobelx", c, o)
            self.rbr_conv1x1_sby_branch = SeqConv3x3("conv1x1-sobely", c, o)
            self.rbr_conv1x1_lpl_branch = SeqConv3x3("conv1x1-laplacian", c, o)

    _BRANCHES = ("rbr_3x3_branch", "rbr_3x1_branch", "rbr_1x3_branch", "rbr_1x1_branch", "rbr_1x1_3x3_branch_1x1",
                 "rbr_1x1_3x3_branch_3x3", "rbr_conv1x1_sbx_branch", "rbr_conv1x1_sby_branch", "rbr_conv1x1_lpl_branch")

    @torch.no_grad()
    def get_equivalent_kernel_bias(self) -> Tuple[Tensor, Tensor]:
        """The single 3x3 kernel/bias equal to the sum of the nine branches (RepRFN.py:210-241)."""
        k = self.rbr_3x3_branch.weight.clone()
        b = self.rbr_3x3_branch.bias.clone()
        k[:, :, 1:2, 1:2] += self.rbr_1x1_branch.weight      # 1x1 at the centre tap
        k[:, :, 1:2, :] += self.rbr_1x3_branch.weight        # 1x3 on the middle row
        k[:, :, :, 1:2] += self.rbr_3x1_branch.weight        # 3x1 on the middle column
        b += self.rbr_1x1_branch.bias + self.rbr_1x3_branch.bias + self.rbr_3x1_branch.bias
        # 1x1 (c -> 2c, no bias) then 3x3 (2c -> o, no bias): contract the middle channels
        k += torch.einsum("omyx,mi->oiyx", self.rbr_1x1_3x3_branch_3x3.weight, self.rbr_1x1_3x3_branch_1x1.weight[:, :, 0, 0])
        idx = torch.arange(min(self.in_channels, self.out_channels), device=k.device)
        k[idx, idx, 1, 1] += 1.0                             # identity branch
        for seq in (self.rbr_conv1x1_sbx_branch, self.rbr_conv1x1_sby_branch, self.rbr_conv1x1_lpl_branch):
            sk, sb = seq.rep_params()
            k += sk
            b += sb
        return k, b

    def equivalent_kernel_bias_autograd(self) -> Tuple[Tensor, Tensor]:
        """get_equivalent_kernel_bias written with out-of-place operations, so that the gradient of the merged
        kernel flows back to the nine branches (the train-time branch fusion: one convolution per block instead
        of nine, RepRFN.py:183-191)."""
        if self.deploy:
            return self.rbr_reparam.weight, self.rbr_reparam.bias
        pad = torch.nn.functional.pad
        k = (self.rbr_3x3_branch.weight + pad(self.rbr_1x1_branch.weight, (1, 1, 1, 1))
             + pad(self.rbr_1x3_branch.weight, (0, 0, 1, 1)) + pad(self.rbr_3x1_branch.weight, (1, 1, 0, 0))
             + torch.einsum("omyx,mi->oiyx", self.rbr_1x1_3x3_branch_3x3.weight, self.rbr_1x1_3x3_branch_1x1.weight[:, :, 0, 0]))
        b = self.rbr_3x3_branch.bias + self.rbr_1x1_branch.bias + self.rbr_1x3_branch.bias + self.rbr_3x1_branch.bias
        k = k + _identity_kernel(self.out_channels, self.in_channels, k.device, k.dtype)
        for seq in (self.rbr_conv1x1_sbx_branch, self.rbr_conv1x1_sby_branch, self.rbr_conv1x1_lpl_branch):
            sk, sb = seq.rep_params()
            k = k + sk
            b = b + sb
        return k, b

    def switch_to_deploy(self) -> None:
        if self.deploy:
            return
        kernel, bias = self.get_equivalent_kernel_bias()
        self.rbr_reparam = nn.Conv2d(self.in_channels, self.out_channels, 3, 1, 1).to(kernel.device)
        self.rbr_reparam.weight.data = kernel
        self.rbr_reparam.bias.data = bias
        for name in self._BRANCHES:
            delattr(self, name)
        self.deploy = True

    def _version_tag(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def kernel_bias(self) -> Tuple[Tensor, Tensor]:
        """Deploy-form kernel of the block, cached until a parameter changes."""
        if self.deploy:
            return self.rbr_reparam.weight.detach(), self.rbr_reparam.bias.detach()
        tag = self._version_tag()
        hit = getattr(self, "_eq_cache", None)
        if hit is None or hit[0] != tag:
            hit = (tag, self.get_equivalent_kernel_bias())
            object.__setattr__(self, "_eq_cache", hit)
        return hit[1]


def merged_kernels_autograd(blocks) -> list:
    """``equivalent_kernel_bias_autograd`` of several RepBlocks of one shape at once: the branch tensors are stacked
    and the weight algebra runs once on the stack, so a training step launches a few dozen small kernels for it instead
    of a few dozen per block.  Returns [(kernel, bias)] in the order of ``blocks``; same operations, same values."""
    blocks = list(blocks)
    if not blocks:
        return []
    shape = (blocks[0].in_channels, blocks[0].out_channels)
    if any(b.deploy for b in blocks) or any((b.in_channels, b.out_channels) != shape for b in blocks) or len(blocks) == 1:
        return [b.equivalent_kernel_bias_autograd() for b in blocks]
    pad = torch.nn.functional.pad

    def st(get):
        return torch.stack([get(b) for b in blocks])

    k = (st(lambda b: b.rbr_3x3_branch.weight) + pad(st(lambda b: b.rbr_1x1_branch.weight), (1, 1, 1, 1))
         + pad(st(lambda b: b.rbr_1x3_branch.weight), (0, 0, 1, 1)) + pad(st(lambda b: b.rbr_3x1_branch.weight), (1, 1, 0, 0))
         + torch.einsum("nomyx,nmi->noiyx", st(lambda b: b.rbr_1x1_3x3_branch_3x3.weight),
                        st(lambda b: b.rbr_1x1_3x3_branch_1x1.weight)[:, :, :, 0, 0]))
    bias = (st(lambda b: b.rbr_3x3_branch.bias) + st(lambda b: b.rbr_1x1_branch.bias)
            + st(lambda b: b.rbr_1x3_branch.bias) + st(lambda b: b.rbr_3x1_branch.bias))
    k = k + _identity_kernel(shape[1], shape[0], k.device, k.dtype)
    for name in ("rbr_conv1x1_sbx_branch", "rbr_conv1x1_sby_branch", "rbr_conv1x1_lpl_branch"):
        # SeqConv3x3.rep_params on the stack
        dw = (st(lambda b: getattr(b, name).scale) * st(lambda b: getattr(b, name).mask))[:, :, 0]       # (N,O,3,3)
        k = k + dw[:, :, None, :, :] * st(lambda b: getattr(b, name).k0)[:, :, :, 0, 0][:, :, :, None, None]
        bias = bias + st(lambda b: getattr(b, name).b0) * dw.sum(dim=(2, 3)) + st(lambda b: getattr(b, name).bias)
    return list(zip(k.unbind(0), bias.unbind(0)))


class ESA(nn.Module):
    def __init__(self, esa_channels: int, n_feats: int) -> None:
        super().__init__()
        f = esa_channels
        self.conv1 = nn.Conv2d(n_feats, f, kernel_size=1)
        self.conv_f = nn.Conv2d(f, f, kernel_size=1)
        self.conv2 = nn.Conv2d(f, f, kernel_size=3, stride=2, padding=0)
        self.conv3 = nn.Conv2d(f, f, kernel_size=3, padding=1)
        self.conv4 = nn.Conv2d(f, n_feats, kernel_size=1)

    def run(self, x: Tensor, n_feats: int) -> Tensor:
        """x (N,H,W,n_feats) bf16 -> x * sigmoid(conv4(resize(conv3(maxpool(conv2(c1)))) + conv_f(c1))), c1 = conv1(x)."""
        f = self.conv1.out_channels
        h, w = x.shape[1:3]
        c1_ = F_.conv2d_auto(x, self.conv1.weight, self.conv1.bias, stride=1, padding=0)      # 1x1 on the MFMA kernel
        if F_.use_conv_gemm:   # 3x3 / stride 2 / pad 0 on the implicit-GEMM kernel: only the strided outputs are computed
            c1 = F_.conv2d_auto(c1_, self.conv2.weight, self.conv2.bias, stride=2, padding=0)
        else:                  # ... or the stride-1 pad-1 layer at the odd pixels 2o + 1 (same sums per output)
            full = F_.conv2d_auto(c1_, self.conv2.weight, self.conv2.bias, stride=1, padding=1)
            ho, wo = (h - 3) // 2 + 1, (w - 3) // 2 + 1
            c1 = full[:, 1:2 * ho:2, 1:2 * wo:2, :].contiguous()
        v_max = F_.max_pool2d_nhwc(c1, 7, 3)
        c3 = F_.conv2d_auto(v_max, self.conv3.weight, self.conv3.bias, stride=1, padding=1)
        if f == 16 and n_feats in (16, 32, 48, 64) and os.environ.get("SRHIP_NO_ESA_TAIL", "0") != "1":
            # conv_f, the resize + add, conv4 and the gate as one pass; weights rounded to bf16 as the MFMA layers' are
            wf, w4 = self._bf16_weights()
            return F_.esa_tail(x, c1_, c3, wf, self.conv_f.bias.detach(), w4, self.conv4.bias.detach(), n_feats)
        cf = F_.conv2d_auto(c1_, self.conv_f.weight, self.conv_f.bias, stride=1, padding=0)
        s = F_.resize_bilinear_nhwc(c3, h, w, addend=cf)
        c4 = F_.conv2d_auto(s, self.conv4.weight, self.conv4.bias, stride=1, padding=0)
        assert c4.shape[-1] == x.shape[-1]
        return F_.sigmoid_gate(x, c4)

    def _bf16_weights(self):
        """conv_f's and conv4's kernels rounded to bf16 (kept as fp32), cached until a parameter changes."""
        tag = tuple((p._version, p.data_ptr()) for p in (self.conv_f.weight, self.conv4.weight))
        hit = getattr(self, "_tail_cache", None)
        if hit is None or hit[0] != tag:
            r = lambda t: t.detach().to(torch.bfloat16).float().reshape(t.shape[0], -1).contiguous()
            hit = (tag, (r(self.conv_f.weight), r(self.conv4.weight)))
            object.__setattr__(self, "_tail_cache", hit)
        return hit[1]


def _esa_train(esa: "ESA", x: Tensor, rec=None) -> Tensor:
    """ESA.run with differentiable operators (x: 64-channel padded feature map).  ``rec(name, tensor)`` records the
    stored map of every operator (tests)."""
    from . import autograd_ops as A
    rec = rec or (lambda name, t: t)
    cs = x.shape[-1]
    c1_ = rec("c1_", A.conv1x1(x, esa.conv1.weight, esa.conv1.bias))
    # 3x3 / stride 2 / pad 0 = the stride-1 pad-1 layer at the odd pixels (2o + 1)
    full = A.conv3x3(c1_, esa.conv2.weight, esa.conv2.bias)
    ho, wo = (full.shape[1] - 3) // 2 + 1, (full.shape[2] - 3) // 2 + 1
    c1 = rec("c1", full[:, 1:2 * ho:2, 1:2 * wo:2, :].contiguous())
    v_max = rec("vmax", A.max_pool2d(c1, 7, 3))
    c3 = rec("c3", A.conv3x3(v_max, esa.conv3.weight, esa.conv3.bias))
    cf = rec("cf", A.conv1x1(c1_, esa.conv_f.weight, esa.conv_f.bias))
    s = rec("s", A.resize_add(c3, cf))
    c4 = rec("c4", A.conv1x1(s, esa.conv4.weight, esa.conv4.bias, cout_stride=cs))
    return rec("out", A.sigmoid_gate(x, c4))


class RepRFB(nn.Module):
    def __init__(self, feature_nums: int, act_type: str = "lrelu", deploy: bool = False) -> None:
        super().__init__()
        self.repblock1 = RepBlock(feature_nums, feature_nums, act_type, deploy)
        self.repblock2 = RepBlock(feature_nums, feature_nums, act_type, deploy)
        self.repblock3 = RepBlock(feature_nums, feature_nums, act_type, deploy)
        self.conv3 = _conv_layer(feature_nums, feature_nums, 3)
        self.esa = ESA(16, feature_nums)


class Upsample_Block(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, upscale_factor: int = 2, kernel_size: int = 3, stride: int = 1) -> None:
        super().__init__()
        self.conv = _conv_layer(in_channels, out_channels * (upscale_factor ** 2), kernel_size, stride)
        self.upscale_factor, self.out_channels = upscale_factor, out_channels


class RepRFN(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, feature_nums: int = 48, sr_scale: int = 4,
                 deploy: bool = False) -> None:
        super().__init__()
        if feature_nums % 16 != 0 or feature_nums > 64:
            raise ValueError("RepRFN on libsrhip: feature_nums must be 16, 32, 48 or 64")
        if out_channels * sr_scale ** 2 > 64:
            raise ValueError("RepRFN on libsrhip: out_channels * sr_scale^2 must not exceed 64")
        self.fea_conv = _conv_layer(in_channels, feature_nums, 3)
        self.reprfb1 = RepRFB(feature_nums, "lrelu", deploy)
        self.reprfb2 = RepRFB(feature_nums, "lrelu", deploy)
        self.reprfb3 = RepRFB(feature_nums, "lrelu", deploy)
        self.reprfb4 = RepRFB(feature_nums, "lrelu", deploy)
        self.lr_conv = _conv_layer(feature_nums, feature_nums, 3)
        self.upsampler = Upsample_Block(feature_nums, out_channels, upscale_factor=sr_scale)
        self.in_channels, self.out_channels, self.feature_nums, self.sr_scale = in_channels, out_channels, feature_nums, sr_scale
        self._packs = {}

    # ------------------------------------------------------------------ packed-weight cache
    def invalidate_packs(self) -> None:
        """Drop every packed-weight copy and every cached merged RepBlock kernel.  The caches are keyed on the
        parameters' version counters, which in-place writes through ``Parameter.data`` do not bump: call this after
        such a write."""
        self._packs.clear()
        for m in self.modules():
            if hasattr(m, "_eq_cache"):
                object.__setattr__(m, "_eq_cache", None)
            if hasattr(m, "_tail_cache"):
                object.__setattr__(m, "_tail_cache", None)
        F_.invalidate_pack_cache()

    def _conv(self, key: str, weight: Tensor, bias: Tensor, x: Tensor, cin: int, **kw) -> Tensor:
        tag = (weight._version, weight.data_ptr(), weight.device)
        hit = self._packs.get(key)
        if hit is None or hit[0] != tag:
            hit = (tag, F_.conv3x3_pack_weights(weight.detach().contiguous(), cin, x.shape[-1]), weight)  # keep the source alive
            self._packs[key] = hit
        return F_.conv3x3(x, hit[1], bias.detach(), weight.shape[0], **kw)

    def forward(self, inputs: Tensor) -> Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self.forward_train(inputs)
        return self.infer(inputs)

    def forward_train(self, inputs: Tensor, trace: Optional[dict] = None) -> Tensor:
        """Differentiable forward (core/trainer.py:158-161 works unchanged): the same graph as infer(), each RepBlock
        as ONE convolution with the kernel merged from its nine branches by differentiable weight algebra, every
        activation-space operator and its backward on the HIP kernels.  Feature maps travel padded to 64 channels.
        ``trace`` (a dict, tests only) receives every map the pass stores, keyed by stage name."""
        from . import autograd_ops as A

        def rec(name, t, prefix=""):
            if trace is not None:
                trace[prefix + name] = t
            return t
        fn = self.feature_nums
        cs = 64 if fn > 32 else F_.pad16(fn)          # channel counts the weight-gradient kernel tiles: 16, 32, 64
        x = F_.nchw_f32_to_nhwc_bf16(inputs.detach().float())
        fea = rec("fea", A.conv3x3(x, self.fea_conv.weight, self.fea_conv.bias, cout_stride=cs))
        out = fea
        merged = merged_kernels_autograd(getattr(getattr(self, f"reprfb{bi}"), f"repblock{ri}")
                                         for bi in range(1, 5) for ri in range(1, 4))
        for bi in range(1, 5):
            rfb = getattr(self, f"reprfb{bi}")
            # repblock1..3 and conv3, each with LeakyReLU, as one chain (RepRFN.py:257-262)
            layers = merged[(bi - 1) * 3:bi * 3] + [(rfb.conv3.weight, rfb.conv3.bias)]
            ch = A.conv3x3_lrelu_chain(out, layers, _SLOPE, cout_stride=cs)
            if trace is not None:     # the chain's stored maps y_1 .. y_4 (its autograd node keeps x, y_1 .. y_n, W_1 .. W_n)
                for li, y_i in enumerate(ch.grad_fn.saved_tensors[1:len(layers) + 1]):
                    trace[f"b{bi}.chain{li + 1}"] = y_i
            t = rec("t", A.add(ch, out), f"b{bi}.")
            out = _esa_train(rfb.esa, t, (lambda name, tt, p=f"b{bi}.": rec(name, tt, p)) if trace is not None else None)
        out = rec("lradd", A.add(rec("lr", A.conv3x3(out, self.lr_conv.weight, self.lr_conv.bias, cout_stride=cs)), fea))
        up = self.upsampler
        c = rec("up", A.conv3x3(out, up.conv.weight, up.conv.bias, cout_stride=64 if up.conv.out_channels > 32 else None))
        return A.pixel_shuffle_out(c, self.out_channels, up.upscale_factor)

    @torch.no_grad()
    def infer(self, inputs: Tensor) -> Tensor:
        fn = self.feature_nums
        x = F_.nchw_f32_to_nhwc_bf16(inputs.float())
        fea = self._conv("fea", self.fea_conv.weight, self.fea_conv.bias, x, self.in_channels)
        out = fea
        for bi in range(1, 5):
            rfb = getattr(self, f"reprfb{bi}")
            t = out
            for ri in range(1, 4):
                k, b = getattr(rfb, f"repblock{ri}").kernel_bias()
                t = self._conv(f"b{bi}r{ri}", k, b, t, fn, act=F_.ACT_LRELU, slope=_SLOPE)
            # inputs + lrelu(conv3(.)), RepRFN.py:261-262
            t = self._conv(f"b{bi}c3", rfb.conv3.weight, rfb.conv3.bias, t, fn, act=F_.ACT_LRELU, slope=_SLOPE, res=out)
            out = rfb.esa.run(t, fn)
        out = self._conv("lr", self.lr_conv.weight, self.lr_conv.bias, out, fn, res=fea)  # RepRFN.py:334-336
        up = self.upsampler
        c = self._conv("up", up.conv.weight, up.conv.bias, out, fn)
        return F_.pixel_shuffle_nhwc_to_nchw_f32(c, self.out_channels, up.upscale_factor)  # RepRFN.py:272-276
