"""Drop-in for the temporal variant ``core.models.model_final.SuperSampler``
(core/models/model_final.py:430-519) on libsrhip.so: same constructor
``SuperSampler(frame_ch, buffer_ch, sr_scale, n_history)``, same positional ``forward(lr_frame,
hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer) -> (sr_frame, enhance_weight)``,
the same module tree (hence identical ``state_dict`` keys/shapes and seed-for-seed initialisation:
``xavier_normal_`` on every ``nn.Conv2d``, PyTorch defaults on the transposed convolutions).

Execution: every convolution, transposed convolution, pixel (un)shuffle, warp, x2 up-sampling and the
gradient map run on the HIP kernels, activations NHWC bf16.  Under autograd (training) the same forward
is assembled from differentiable operators over the same launches (``autograd_ops.conv2d``: data
gradient = the operator on transposed / flipped weights, stride 2 through the zero-inserted gradient;
weight gradient = the MFMA weight-gradient kernel per <= 64-channel slice pair, a GEMM for 1x1 layers);
the inputs receive no gradient and the history features are detached, as in the reference
(model_final.py:489).  Every convolution goes to the MFMA kernel through ``functional.conv2d_auto`` (1x1, stride-2 and
wide layers included; transposed 2x2 convolutions as 1x1 layers plus a depth-to-space copy).  The
disocclusion masks with the marked history and auxiliary planes (model_final.py:53-139,472-486,501-504)
are ONE fused kernel (``functional.disocclusion``), the depth clamp (:40-46) ``functional.clamp_depth``
and the gated sum of the main stage (:422) ``functional.attention_fuse``.  The warp is
``tools_math.backward_warping`` (the reference's own helper raises for every input; see that module).
"""
import os
from typing import Tuple

import torch
from torch import Tensor, nn

from . import autograd_ops as A_
from . import functional as F_
from .tools_math import backward_warping_history


def _add(a: Tensor, b: Tensor) -> Tensor:
    return A_.add(a, b) if torch.is_grad_enabled() else F_.add_bf16(a, b)


# ------------------------------------------------------------------------------------------------
# Recording of the stored maps of one pass (tests: a bf16-storage oracle's backward is run on the very maps this pass
# stored, so that both take the same ReLU decisions).  Keys: "<module path>#<k-th call in the pass>" for a layer's
# output, a few fixed names for the rest.  Values: (tensor, number of real channels) for NHWC bf16 maps, (tensor, None)
# for fp32 NCHW tensors.
# ------------------------------------------------------------------------------------------------
_TRACE = None


class trace_maps:
    def __init__(self, model: nn.Module, store: dict) -> None:
        self.state = {"names": {id(m): n for n, m in model.named_modules()}, "count": {}, "maps": store}

    def __enter__(self):
        global _TRACE
        _TRACE = self.state
        return self.state["maps"]

    def __exit__(self, *exc):
        global _TRACE
        _TRACE = None
        return False


def _rec(where, t: Tensor, channels=None, suffix: str = "") -> Tensor:
    if _TRACE is not None:
        if isinstance(where, str):
            key = where
        else:
            base = _TRACE["names"][id(where)] + suffix
            k = _TRACE["count"].get(base, 0)
            _TRACE["count"][base] = k + 1
            key = f"{base}#{k}"
        _TRACE["maps"][key] = (t.detach(), channels)
    return t


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's module tree
# ------------------------------------------------------------------------------------------------
class GradientMap(nn.Module):
    def __init__(self, channels=None, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        kv = torch.tensor([[-1, -1, -1], [0, 0, 0], [1, 1, 1]], dtype=torch.float32).view(1, 1, 3, 3)
        kh = torch.tensor([[-1, 0, 1], [-1, 0, 1], [-1, 0, 1]], dtype=torch.float32).view(1, 1, 3, 3)
        if channels is not None:
            kv, kh = kv.repeat(channels, 1, 1, 1), kh.repeat(channels, 1, 1, 1)
        self.register_buffer("kernel_v", kv)
        self.register_buffer("kernel_h", kh)


class ConvActivation(nn.Sequential):
    def __init__(self, cin: int, cout: int, kernel_size: int = 3, stride: int = 1, padding: int = 1) -> None:
        super().__init__(nn.Conv2d(cin, cout, kernel_size, stride, padding), nn.ReLU(inplace=True))

    def run(self, x: Tensor, res: Tensor = None) -> Tensor:
        c = self[0]
        if torch.is_grad_enabled():   # training graph: the same launch as a differentiable operator
            y = _rec(self, A_.conv2d(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0], act=F_.ACT_RELU), c.out_channels)
            return y if res is None else A_.add(y, res)
        if _TRACE is not None and res is not None:   # recorded passes keep the layer's own output apart from the sum
            y = _rec(self, F_.conv2d_auto(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0], act=F_.ACT_RELU), c.out_channels)
            return F_.add_bf16(y, res)
        return _rec(self, F_.conv2d_auto(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0], act=F_.ACT_RELU, res=res),
                    c.out_channels)


class TransConvActivation(nn.Sequential):
    def __init__(self, cin: int, cout: int, kernel_size: int = 2, stride: int = 2, padding: int = 0) -> None:
        super().__init__(nn.ConvTranspose2d(cin, cout, kernel_size, stride, padding), nn.ReLU(inplace=True))

    def run(self, x: Tensor) -> Tensor:
        c = self[0]
        if torch.is_grad_enabled():
            # stride = kernel size: a 1x1 layer to 4 * cout channels (phase-major) and a depth-to-space rearrangement; the
            # weight rearrangement and the depth-to-space are differentiated by torch, the layer by its own operator
            cin, cout = c.weight.shape[0], c.weight.shape[1]
            w1 = c.weight.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)
            y = A_.conv2d(x, w1, c.bias.repeat(4), stride=1, padding=0, act=F_.ACT_RELU)
            n, h, w, _ = x.shape
            return _rec(self, y[..., :4 * cout].reshape(n, h, w, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, cout).contiguous(), cout)
        return _rec(self, F_.conv_transpose2x2s2_auto(x, c.weight, c.bias, act=F_.ACT_RELU), c.weight.shape[1])


def _run_layers(mods, x):
    for m in mods:
        x = m.run(x)
    return x


_CHAIN_TRAIN = os.environ.get("SRHIP_NO_TRAIN_CHAIN", "0") != "1"   # A/B switch of the chained training nodes


def _run_seq(seq, x: Tensor, addend: Tensor = None) -> Tensor:
    """The layers of `seq` in order; `addend` (a skip connection, model_final.py:308-313) is added to the result: by the
    up-sampling pass itself where the sequence ends in one and no graph is recorded, by an add otherwise."""
    last = len(seq) - 1
    mods = list(seq)
    skip = 0
    for i, m in enumerate(mods):
        if skip:
            skip -= 1
            continue
        if isinstance(m, ConvActivation) and torch.is_grad_enabled() and _TRACE is None and _CHAIN_TRAIN:
            # training graph: a run of conv + ReLU layers whose intermediate maps nobody else reads is ONE autograd node
            # (each intermediate ReLU backward rides on the data-gradient launch behind it); recorded passes keep the layers apart
            run = [m]
            while i + len(run) < len(mods) and isinstance(mods[i + len(run)], ConvActivation):
                run.append(mods[i + len(run)])
            if len(run) >= 2:
                x = A_.conv_relu_chain(x, [(r[0].weight, r[0].bias, r[0].stride[0], r[0].padding[0]) for r in run])
                skip = len(run) - 1
                continue
        if isinstance(m, (ConvActivation, TransConvActivation)):
            x = m.run(x)
        elif isinstance(m, nn.Upsample):  # x2 bilinear, align_corners=True (model_final.py:281-289)
            if torch.is_grad_enabled():   # training graph: the same kernel with its adjoint as backward
                x = A_.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2], align_corners=True)
            else:
                fused = addend if i == last else None
                x = F_.resize_bilinear_nhwc(x, 2 * x.shape[1], 2 * x.shape[2], addend=fused, align_corners=True)
                if fused is not None:
                    addend = None
        else:
            raise TypeError(f"unexpected module {type(m).__name__}")
    return x if addend is None else _add(x, addend)


class FrameEncoder(nn.Module):
    def __init__(self, cin: int, cout: int, inner: int = 24, return_head_feature: bool = False) -> None:
        super().__init__()
        self.conv_head = ConvActivation(cin, cout)
        self.conv_main = nn.Sequential(ConvActivation(cout, cout), ConvActivation(cout, inner),
                                       ConvActivation(inner, cout), ConvActivation(cout, cout))
        self.return_head_feature = return_head_feature

    def run(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        head = self.conv_head.run(x)
        return _run_seq(self.conv_main, head), head


class GBuffersEncoder(nn.Module):
    def __init__(self, cin: int, cout: int, factor: int = 4) -> None:
        super().__init__()
        inner = cin * factor ** 2
        self.factor = factor
        self.fold = nn.PixelUnshuffle(factor)
        self.conv_main = nn.Sequential(ConvActivation(inner, inner), ConvActivation(inner, inner, 1, padding=0),
                                       ConvActivation(inner, inner, 1, padding=0), ConvActivation(inner, cout, 1, padding=0))
        e = 7 * factor ** 2
        self.conv_enhance = nn.Sequential(ConvActivation(e, e), ConvActivation(e, cout, 1, padding=0),
                                          ConvActivation(cout, cout, 1, padding=0), ConvActivation(cout, cout, 1, padding=0))
        self.enhance_param = nn.Parameter(torch.tensor(0.45), requires_grad=True)

    def run(self, buffer: Tensor, want_max: bool = False):
        """buffer (B,10,H,W) fp32 -> (B,h,w,cout) bf16 (model_final.py:250-258); want_max: also buffer.max() (0-d device
        tensor), found by the pass that folds the buffer."""
        out = self._run(buffer, want_max)
        return out if want_max else out[0]

    def _run(self, buffer: Tensor, want_max: bool):
        r2 = self.factor ** 2
        # folded buffer, channel c*r2 + i*r + j, straight in the convolutions' layout; below: normal (0:3), base colour
        # (6:9) and depth (9:10) planes of it
        if want_max:
            fn, vmax = F_.pixel_unshuffle_to_nhwc_bf16(buffer, self.factor, want_max=True)
        else:
            fn, vmax = F_.pixel_unshuffle_to_nhwc_bf16(buffer, self.factor), None
        # the three 1x1 layers behind each 3x3 layer as one pass (inference; recorded passes keep every layer's map)
        chain = not torch.is_grad_enabled() and _TRACE is None
        tail = lambda seq: [(m[0].weight, m[0].bias, F_.ACT_RELU) for m in list(seq)[1:]]
        chans = lambda seq: [seq[1][0].in_channels] + [m[0].out_channels for m in list(seq)[1:]]
        if chain and len(self.conv_main) == 4 and F_.conv1x1_chain_supported(chans(self.conv_main), False):
            fmap = F_.conv1x1_chain(self.conv_main[0].run(fn), tail(self.conv_main))
        else:
            fmap = _run_seq(self.conv_main, fn)
        # picked from the converted map (the conversion rounds element by element, so the order does not matter)
        if not torch.is_grad_enabled() and r2 % 16 == 0:
            e = ((fn, 0, 3 * r2), (fn, 6 * r2, 4 * r2))   # the first enhance layer reads the two channel windows in place
        elif r2 % 2 == 0 and fn.shape[-1] % 2 == 0:      # copy channel pairs as 32-bit words
            f32 = fn.view(torch.int32)
            e = torch.cat((f32[..., 0:3 * r2 // 2], f32[..., 3 * r2:5 * r2]), dim=-1).view(torch.bfloat16)
        else:
            e = torch.cat((fn[..., 0:3 * r2], fn[..., 6 * r2:10 * r2]), dim=-1)
        if chain and len(self.conv_enhance) == 4 and F_.conv1x1_chain_supported(chans(self.conv_enhance), True):
            # ... and fmap + enhance_param * (the chain's result) on its last epilogue
            return F_.conv1x1_chain(self.conv_enhance[0].run(e), tail(self.conv_enhance), alpha=self._enhance_scale(), res=fmap), vmax
        e = _run_seq(list(self.conv_enhance)[:-1], e) if (torch.is_grad_enabled() and _TRACE is None and _CHAIN_TRAIN) \
            else _run_layers(list(self.conv_enhance)[:-1], e)
        last = self.conv_enhance[-1][0]
        if torch.is_grad_enabled():   # enhance_param is trained: the scale and the add stay visible to autograd
            y = _rec(self.conv_enhance[-1], A_.conv2d(e, last.weight, last.bias, stride=1, padding=0, act=F_.ACT_RELU), last.out_channels)
            # the scale stays fp32 (inference passes float(enhance_param) as the epilogue's alpha and rounds the sum
            # once): the product and the sum are formed in fp32, and enhance_param's gradient is an fp32 reduction
            if self.enhance_param.dtype == torch.float32 and fmap.shape == y.shape and os.environ.get("SRHIP_NO_SCALE_ADD", "0") != "1":
                return _rec("buffers_encoder.out", A_.scale_add(fmap, y, self.enhance_param), fmap.shape[-1]), vmax   # the same, one pass
            return _rec("buffers_encoder.out", (fmap.float() + self.enhance_param * y.float()).to(torch.bfloat16), fmap.shape[-1]), vmax
        # fmap + enhance_param * relu(conv(e)): scale and add ride on the last convolution's epilogue
        return F_.conv2d_auto(e, last.weight, last.bias, stride=1, padding=0, act=F_.ACT_RELU,
                              alpha=self._enhance_scale(), res=fmap), vmax

    def _enhance_scale(self) -> float:
        """enhance_param as a host number, read from the device only when the parameter changed (reading it per
        frame would wait for the launch queue to drain)."""
        p = self.enhance_param
        tag = (p._version, p.data_ptr())
        hit = getattr(self, "_enhance_cache", None)
        if hit is None or hit[0] != tag:
            hit = (tag, float(p))
            object.__setattr__(self, "_enhance_cache", hit)
        return hit[1]


class HistoryRefiner(nn.Module):
    def __init__(self, cin: int = 64, cout: int = 32, aux_channels: int = 16, down_chs=(32, 48, 64, 64),
                 up_chs=(32, 32, 48, 64)) -> None:
        super().__init__()
        assert down_chs[0] == up_chs[0], "Output channels must be equal to the sum of the last two channels in the encoder."
        self.refine_in = ConvActivation(cin, down_chs[0])
        self.conv0 = ConvActivation(down_chs[0] + aux_channels, down_chs[0])
        self.down0 = ConvActivation(down_chs[0], down_chs[1], 3, stride=2)
        self.conv1 = ConvActivation(down_chs[1], down_chs[1])
        self.down1 = ConvActivation(down_chs[1], down_chs[2], 3, stride=2)
        self.conv2 = ConvActivation(down_chs[2], down_chs[2])
        self.down2 = ConvActivation(down_chs[2], down_chs[3], 3, stride=2)
        self.conv3 = ConvActivation(down_chs[3], down_chs[3])
        up = lambda a, b: nn.Sequential(ConvActivation(a, b), nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))
        self.up3 = up(down_chs[3], up_chs[3])
        self.up2 = up(up_chs[3], up_chs[2])
        self.up1 = up(up_chs[2], up_chs[1])
        self.up0 = ConvActivation(up_chs[1], up_chs[0])
        self.refine_out = ConvActivation(up_chs[0], cout, 1, padding=0)

    def run(self, x: Tensor, aux: Tensor, motion: Tensor) -> Tensor:
        """x (N,h,w,64) bf16; aux (N,14,h,w), motion (N,2,h,w) fp32 (model_final.py:295-315).  aux may come with two spare
        planes behind its fourteen ((1,16,h,w), ``functional.disocclusion(aux_planes=16)``): |motion| is written there."""
        x_in = self.refine_in.run(x)
        if aux.shape[1] == 16:
            torch.abs(motion, out=aux[:, 14:16])
            extra = F_.nchw_f32_to_nhwc_bf16(aux)
        else:
            extra = F_.nchw_f32_to_nhwc_bf16(torch.cat((aux, motion.abs()), dim=1).contiguous())   # 16 channels
        x0 = self.conv0.run((x_in, extra))                # the layer reads the concatenation in place
        x1 = self.conv1.run(self.down0.run(x0))
        x2 = self.conv2.run(self.down1.run(x1))
        x3 = self.conv3.run(self.down2.run(x2))
        s2 = _run_seq(self.up3, x3, addend=x2)            # up3(x3) + x2, ...
        s1 = _run_seq(self.up2, s2, addend=x1)
        s0 = _run_seq(self.up1, s1, addend=x0)
        return self.refine_out.run(self.up0.run(s0, res=x_in))


class FusionNetwork(nn.Module):
    def __init__(self, cin: int, first_channel=None, down_chs=(64, 128, 192, 256), up_chs=(64, 64, 128, 192)) -> None:
        super().__init__()
        first_channel = down_chs[0] if first_channel is None else first_channel
        two = lambda a, b, c: nn.Sequential(ConvActivation(a, b), ConvActivation(b, c))
        self.conv0 = two(cin, first_channel, down_chs[0])
        self.down1 = ConvActivation(down_chs[0], down_chs[1], 3, stride=2)
        self.conv1 = two(down_chs[1], down_chs[1], down_chs[1])
        self.down2 = ConvActivation(down_chs[1], down_chs[2], 3, stride=2)
        self.conv2 = two(down_chs[2], down_chs[2], down_chs[2])
        self.down3 = ConvActivation(down_chs[2], down_chs[3], 3, stride=2)
        self.conv3 = two(down_chs[3], down_chs[3], down_chs[3])
        self.up3 = nn.Sequential(ConvActivation(down_chs[3], up_chs[3]), ConvActivation(up_chs[3], up_chs[3]),
                                 TransConvActivation(up_chs[3], up_chs[3], 2, 2))
        self.up2 = nn.Sequential(ConvActivation(up_chs[3] + down_chs[2], 2 * up_chs[2]), ConvActivation(2 * up_chs[2], up_chs[2]),
                                 TransConvActivation(up_chs[2], up_chs[2], 2, 2))
        self.up1 = nn.Sequential(ConvActivation(up_chs[2] + down_chs[1], 2 * up_chs[1]), ConvActivation(2 * up_chs[1], up_chs[1]),
                                 TransConvActivation(up_chs[1], up_chs[1], 2, 2))
        self.up0 = two(up_chs[1] + down_chs[0], up_chs[0], up_chs[0])

    def run(self, x: Tensor) -> Tensor:
        f0 = _run_seq(self.conv0, x)
        f1 = _run_seq(self.conv1, self.down1.run(f0))
        f2 = _run_seq(self.conv2, self.down2.run(f1))
        f3 = _run_seq(self.conv3, self.down3.run(f2))
        f3 = _run_seq(self.up3, f3)
        f2 = _run_seq(self.up2, (f3, f2))                 # skip connections: read in place, never concatenated
        f1 = _run_seq(self.up1, (f2, f1))
        return _run_seq(self.up0, (f1, f0))


class Attention(nn.Module):
    def __init__(self, cin: int, cout=None) -> None:
        super().__init__()
        cout = cin if cout is None else cout
        self.conv = ConvActivation(cin, cout)
        self.w_self = ConvActivation(cin, 1)
        self.w_cross = ConvActivation(cin, 1)

    def run(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        if torch.is_grad_enabled():   # training: the whole padded gate maps (channel 0 counts; the gated sum's backward
            return self.conv.run(x), self.w_self.run(x), self.w_cross.run(x)   # returns the gradient of the whole map)
        if _TRACE is not None:
            return self.conv.run(x), self.w_self.run(x)[..., 0:1], self.w_cross.run(x)[..., 0:1]
        # inference: the two one-channel layers read the same map -> one launch with both kernels (channels 0 and 1)
        w2, b2 = self._gate_kernels()
        gates = F_.conv2d_auto(x, w2, b2, stride=1, padding=1, act=F_.ACT_RELU)
        return self.conv.run(x), gates[..., 0:1], gates[..., 1:2]

    def _gate_kernels(self):
        """w_self's and w_cross's kernels and biases stacked (2, cin, 3, 3) / (2,), cached until a parameter changes."""
        ps = (self.w_self[0].weight, self.w_cross[0].weight, self.w_self[0].bias, self.w_cross[0].bias)
        tag = tuple((p._version, p.data_ptr()) for p in ps)
        hit = getattr(self, "_gate_cache", None)
        if hit is None or hit[0] != tag:
            hit = (tag, (torch.cat((ps[0].detach(), ps[1].detach()), dim=0).contiguous(), torch.cat((ps[2].detach(), ps[3].detach())).contiguous()))
            object.__setattr__(self, "_gate_cache", hit)
        return hit[1]


class MainNetwork(nn.Module):
    def __init__(self, basic_channels: int = 64, curr_channels: int = 64, hist_channels: int = 32,
                 gbuffer_channels: int = 64, sr_scale: int = 4) -> None:
        super().__init__()
        assert curr_channels == hist_channels, "curr_channels and hist_channels must be the same"
        self.sr_scale = sr_scale
        self.grad_map = nn.Sequential(GradientMap(6), nn.PixelUnshuffle(sr_scale))
        self.attn_grad = Attention(6 * sr_scale ** 2, basic_channels)
        self.attn_fmap = Attention(basic_channels, basic_channels)
        self.main_network = FusionNetwork(basic_channels + hist_channels + gbuffer_channels, basic_channels + gbuffer_channels)

    def run(self, fmap_current: Tensor, fmap_history: Tensor, fmap_gbuffer: Tensor, hr_buffer: Tensor) -> Tensor:
        # normal and base-colour planes (model_final.py:417); each range is contiguous per sample, so no copy is made
        if self.sr_scale == 4:   # gradient map, PixelUnshuffle and the layout conversion in one pass over the six planes
            folded = F_.gradient_map_unshuffle_nhwc(hr_buffer, (0, 1, 2, 6, 7, 8), self.grad_map[0].eps)
        else:
            nb, _, hh, ww = hr_buffer.shape
            gfull = torch.empty((nb, 6, hh, ww), dtype=torch.float32, device=hr_buffer.device)
            for i in range(nb):
                F_.gradient_map(hr_buffer[i:i + 1, 0:3], self.grad_map[0].eps, out=gfull[i:i + 1, 0:3])
                F_.gradient_map(hr_buffer[i:i + 1, 6:9], self.grad_map[0].eps, out=gfull[i:i + 1, 3:6])
            folded = F_.pixel_unshuffle_to_nhwc_bf16(gfull, self.sr_scale)
        grad, g_self, g_cross = self.attn_grad.run(folded)
        fmap, f_self, f_cross = self.attn_fmap.run(fmap_current)
        _rec("main_network.grad_map", folded, 6 * self.sr_scale ** 2)
        if torch.is_grad_enabled():
            # model_final.py:422 (gates broadcast over the channels): one launch, formed in fp32 and rounded once, and one
            # launch for its backward (SRHIP_NO_FUSED_GATE=1: the same sum in torch operators)
            if os.environ.get("SRHIP_NO_FUSED_GATE", "0") == "1":
                f_self, g_cross, g_self, f_cross = (t[..., 0:1] for t in (f_self, g_cross, g_self, f_cross))
                cur = (fmap_current.float() + torch.sigmoid(f_self.float() * g_cross.float()) * fmap.float()
                       + torch.sigmoid(g_self.float() * f_cross.float()) * grad.float()).to(torch.bfloat16)
            else:
                cur = A_.attention_fuse(fmap_current, fmap, grad, f_self, g_cross, g_self, f_cross)
        else:                         # ... as one kernel
            cur = F_.attention_fuse(fmap_current, fmap, grad, f_self, g_cross, g_self, f_cross)
        _rec("main_network.cur", cur, cur.shape[-1])
        return self.main_network.run((cur, fmap_history, fmap_gbuffer))


# ------------------------------------------------------------------------------------------------
class SuperSampler(nn.Module):
    def __init__(self, frame_ch: int, buffer_ch: int, sr_scale: int, n_history: int) -> None:
        super().__init__()
        self.n_history = n_history
        fr, bf = 64, 64
        hs = fr // n_history
        dw, up = [64, 128, 192, 256], [64, 64, 128, 192]
        self.frame_encoder = FrameEncoder(frame_ch, fr, return_head_feature=True)
        self.buffers_encoder = GBuffersEncoder(buffer_ch, bf, factor=sr_scale)
        self.history_refiner = HistoryRefiner(fr, hs, aux_channels=(3 + 1 + 3) * 2 + 2)
        self.main_network = MainNetwork(hist_channels=hs * n_history, sr_scale=sr_scale)
        self.conv_out = nn.Sequential(nn.Conv2d(up[0] + dw[0], frame_ch * sr_scale ** 2, 3, 1, 1), nn.PixelShuffle(sr_scale),
                                      nn.Conv2d(frame_ch, frame_ch, 3, 1, 1), nn.ReLU(inplace=True))
        self.sr_scale, self.frame_ch = sr_scale, frame_ch
        self.reset_parameters()

    def reset_parameters(self) -> None:
        for m in self.modules():  # model_final.py:507-519 (transposed convolutions keep their default init)
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def invalidate_packs(self) -> None:
        """Drop every packed-weight copy (keyed on the parameters' version counters, which in-place writes through
        ``Parameter.data`` do not bump)."""
        F_.invalidate_pack_cache()
        object.__setattr__(self.buffers_encoder, "_enhance_cache", None)

    def forward(self, lr_frame: Tensor, hr_buffer: Tensor, lr_buffer: Tensor, motion_vectors: Tensor,
                history_frames: Tensor, history_buffer: Tensor) -> Tuple[Tensor, Tensor]:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training graph: every layer a differentiable operator over the same launches (autograd_ops.conv2d); the
            # inputs receive no gradient, and the history features are detached as in the reference (model_final.py:489)
            return self._forward(lr_frame, hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer)
        return self.infer(lr_frame, hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer)

    @torch.no_grad()
    def infer(self, lr_frame, hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer):
        return self._forward(lr_frame, hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer)

    def _forward(self, lr_frame, hr_buffer, lr_buffer, motion_vectors, history_frames, history_buffer):
        training = torch.is_grad_enabled()
        with torch.no_grad():
            # depth clamp, in place like the reference (model_final.py:462-463)
            for planes in (hr_buffer[:, -1], lr_buffer[:, -1], history_buffer[:, :, -1]):
                if planes.is_contiguous() and planes.dtype == torch.float32:      # one sample: the planes are one range
                    F_.clamp_depth(planes, out=planes)
                else:
                    planes.copy_(F_.clamp_depth(planes))

        # the maximum of the clamped buffer (model_final.py:463) comes out of the pass that folds it; it is checked at
        # the end of the pass (below): reading it here would drain the launch queue
        fmap_bf, depth_max = self.buffers_encoder.run(hr_buffer.contiguous(), want_max=True)
        fmap_current, fmap_shadow = self.frame_encoder.run(F_.nchw_f32_to_nhwc_bf16(lr_frame.float().contiguous()))

        b, f, _, h, w = history_frames.shape
        with torch.no_grad():   # functions of the inputs alone; the history features are detached (model_final.py:489)
            warped_frames, warped_buffer, motion_vectors = backward_warping_history(motion_vectors, history_frames, history_buffer)
            # masks, marked history, auxiliary planes and the loss weight in one kernel (model_final.py:472-486,501-504)
            marked_frames, aux_buffer, enhance_weight = F_.disocclusion(lr_frame.float(), lr_buffer.float(), warped_frames.float(),
                                                                        warped_buffer.float(), aux_planes=16 if b * f == 1 else 14)
            aux_motion = motion_vectors.reshape(b * f, -1, h, w)
            _rec("pre.marked_frames", marked_frames)
            if _TRACE is not None:   # (the copy is for the recorder alone: 29 MB per 544 x 960 frame)
                _rec("pre.aux", aux_buffer[:, :14].clone())
            _rec("pre.motion", aux_motion)
            _rec("pre.enhance_weight", enhance_weight)
            fmap_history = self.frame_encoder.run(F_.nchw_f32_to_nhwc_bf16(marked_frames))[0]
        fmap_history = self.history_refiner.run(fmap_history, aux_buffer, aux_motion)          # (b*f, h, w, hs)
        if f > 1:  # (b, f*hs, h, w) in the reference: history index major, then channels
            fmap_history = fmap_history.view(b, f, h, w, -1).permute(0, 2, 3, 1, 4).reshape(b, h, w, -1).contiguous()

        fmap = self.main_network.run(fmap_current, fmap_history, fmap_bf, hr_buffer)
        c0, c2 = self.conv_out[0], self.conv_out[2]
        if training:
            t = _rec("conv_out.0", A_.conv2d((fmap, fmap_shadow), c0.weight, c0.bias), c0.out_channels)
            if self.sr_scale == 4 and self.frame_ch <= 4 and _TRACE is None:
                # PixelShuffle straight into the last layer's padded input layout, and the output's layout change as one
                # pass (each with a one-pass backward) instead of permute / pad / cast chains on the 16-channel HR image
                x16 = A_.pixel_shuffle4_to_nhwc16(t, self.frame_ch)
                y = A_.conv2d(x16, c2.weight, c2.bias, act=F_.ACT_RELU)
                sr_frame = A_.nhwc_to_nchw_f32(y, self.frame_ch)
            else:
                hr = A_.pixel_shuffle_out(t, self.frame_ch, self.sr_scale)                     # (b, 3, H, W) fp32
                x16 = nn.functional.pad(hr.permute(0, 2, 3, 1), (0, 16 - self.frame_ch)).to(torch.bfloat16).contiguous()
                y = _rec("conv_out.2", A_.conv2d(x16, c2.weight, c2.bias, act=F_.ACT_RELU), self.frame_ch)
                sr_frame = y[..., :self.frame_ch].permute(0, 3, 1, 2).float()
            assert depth_max < 10, "Depth value is too large"
            return sr_frame, enhance_weight
        t = F_.conv2d_auto((fmap, fmap_shadow), c0.weight, c0.bias)
        if self.sr_scale == 4 and self.frame_ch == 3:   # PixelShuffle, the last 3 -> 3 layer and its ReLU in one pass
            sr_frame = F_.pixel_shuffle4_conv3x3_c3(t, c2.weight.detach().float(), c2.bias.detach().float(), act=F_.ACT_RELU)
        else:
            if self.sr_scale == 4 and self.frame_ch <= 4:   # PixelShuffle straight into the last layer's padded input layout
                hr16 = F_.pixel_shuffle4_nhwc_to_nhwc16(t, self.frame_ch)
            else:
                hr16 = F_.nchw_f32_to_nhwc_bf16(F_.pixel_shuffle_nhwc_to_nchw_f32(t, self.frame_ch, self.sr_scale))
            wp = F_.conv3x3_pack_weights(c2.weight.detach(), self.frame_ch, 16)
            sr_frame = F_.conv3x3(hr16, wp, c2.bias.detach(), self.frame_ch, act=F_.ACT_RELU, out_nchw_f32=True)

        # model_final.py:464; the comparison waits for the device, so it runs once every launch of the pass is queued
        assert depth_max < 10, "Depth value is too large"
        return sr_frame, enhance_weight
