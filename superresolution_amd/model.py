"""Drop-in for the reference's default network, ``core.models.model.SuperSampler``
(core/models/model.py:98-175), running on libsrhip.so.

Contract kept (core/models/model.py:99,147-175; core/trainer.py:32-35,232,254):
  * ctor ``SuperSampler(frame_channels=3, buffer_channels=10, n_history=1)``;
  * ``forward(batch)`` pops ``lr_frames`` (B,L,3,sh,sw) / ``hr_buffers`` (B,L,10,th,tw), uses the
    last frame of each, sets ``batch["sr_frames"]`` (B,3,th,tw) fp32 and returns the dict;
  * identical ``state_dict`` keys / OIHW fp32 shapes and the same seed-for-seed initialisation
    (xavier_normal_ weights, zero biases, modules built in the same order).

Inside, activations are NHWC bf16, every convolution is one launch of the MFMA kernel with its
bias / ReLU / residual / 0.2-scale fused, and both channel concats are folded into the consumer
convolution (no concat tensor is materialised).
"""
import os
from typing import Dict, List, Union

import torch
from torch import Tensor, nn

from . import functional as F_


def _conv_act(cin: int, cout: int) -> nn.Sequential:
    # parameter container with the reference's key layout "<name>.0.{weight,bias}"
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.ReLU(inplace=True))


class _Residual(nn.Sequential):
    """x + seq(x) in the reference (model.py:17-19); here only a parameter container."""


class _SRResidualBlock(nn.Module):
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.blks = nn.Sequential(nn.Conv2d(channels, channels, 3, 1, 1), nn.ReLU(inplace=True),
                                  nn.Conv2d(channels, channels, 3, 1, 1))


class _FrameExtractor(nn.Module):
    def __init__(self, cin: int, widths: List[int]) -> None:
        super().__init__()
        blks, cur = [], cin
        for wd in widths:
            blks.append(_conv_act(cur, wd))
            cur = wd
        self.blks = nn.ModuleList(blks)


class _GBuffersExtractor(nn.Module):
    def __init__(self, cin: int, widths: List[int]) -> None:
        super().__init__()
        blks, cur = [], cin
        for wd in widths:
            blks += [_conv_act(cur, wd), _Residual(_conv_act(wd, wd), _conv_act(wd, wd))]
            cur = wd
        self.blks = nn.Sequential(*blks)


class _MainNetwork(nn.Module):
    def __init__(self, cin: int, ch: int, n_layers: int) -> None:
        super().__init__()
        self.conv_in = _conv_act(cin, ch)
        self.conv_main = nn.ModuleList([_SRResidualBlock(ch) for _ in range(n_layers)])


class SuperSampler(nn.Module):
    FR_CH, BF_CH, SR_CH = 32, 64, 64

    def __init__(self, frame_channels: int = 3, buffer_channels: int = 10, n_history: int = 1) -> None:
        super().__init__()
        fr, bf, sr = self.FR_CH, self.BF_CH, self.SR_CH
        self.frame_extractor = _FrameExtractor(frame_channels, [fr, fr, fr])
        self.gbuffers_extractor = _GBuffersExtractor(buffer_channels, [bf, bf, bf])
        self.main = _MainNetwork(fr + bf, sr, 6)
        self.after_main = _conv_act(sr + fr, sr)
        self.final_conv = nn.Sequential(nn.Conv2d(sr, sr, 3, 1, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(sr, frame_channels, 3, 1, 1))
        self.n_history = n_history
        self.frame_channels = frame_channels
        self.buffer_channels = buffer_channels
        self._init_parameters()
        self._packs: Dict[str, tuple] = {}
        self.use_pair_kernel = os.environ.get("SRHIP_NO_PAIR", "0") != "1"

    def _init_parameters(self) -> None:
        # same rule and same visiting order as core/models/model.py:120-132
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ packed-weight cache
    def _cached_pack(self, key: str, w: Tensor, build) -> Tensor:
        tag = (w._version, w.data_ptr(), w.device)
        hit = self._packs.get(key)
        if hit is None or hit[0] != tag:
            hit = (tag, build())
            self._packs[key] = hit
        return hit[1]

    def invalidate_packs(self) -> None:
        """Drop every packed-weight copy.  The cache is keyed on each weight's version counter, which in-place writes
        through ``Parameter.data`` (some EMA / clamping code) do not bump: call this after such a write."""
        self._packs.clear()

    def _pack(self, name: str, conv: nn.Conv2d, cin_a: int, ca_pad: int, cin_b: int = 0, cb_pad: int = 0) -> Tensor:
        w = conv.weight
        return self._cached_pack(name, w, lambda: F_.conv3x3_pack_weights(w.detach(), cin_a, ca_pad, cin_b, cb_pad))

    def _dgrad(self, name: str, conv: nn.Conv2d, dz: Tensor, lo: int = 0, n: int = None, mask: Tensor = None,
               res: Tensor = None, alpha: float = 1.0, signbits: Tensor = None, aux_mode: int = F_.AUX_NONE):
        """Gradient w.r.t. input channels [lo, lo+n) of ``conv`` from dz, the gradient of its pre-activation
        output: alpha * conv_transpose(dz), then either masked by a saved ReLU output (``mask``) or added to
        ``res``.  Runs the forward kernel on transposed, flipped weights.  With ``signbits`` (the 1-bit-per-element
        ReLU pattern a forward launch emitted): ``AUX_MASKBITS`` masks the sum with ``res``, ``AUX_DUALBITS`` returns
        the gradient and its masked copy."""
        w = conv.weight
        n = conv.in_channels - lo if n is None else n
        wp = self._cached_pack(f"{name}^T{lo}", w, lambda: F_.conv3x3_pack_weights(
            w.detach(), conv.out_channels, dz.shape[-1], cout=n, cout_off=lo, transpose_flip=True))
        if aux_mode != F_.AUX_NONE:
            return F_.conv3x3(dz, wp, None, n, res=res, act=F_.ACT_NONE, alpha=alpha, signbits=signbits, aux_mode=aux_mode)
        second = mask if mask is not None else res
        return F_.conv3x3(dz, wp, None, n, res=second, act=F_.ACT_NONE, alpha=alpha,
                          res_mode=F_.RES_MASK if mask is not None else F_.RES_ADD)

    def _conv(self, name: str, conv: nn.Conv2d, x: Tensor, cin_a: int, xb: Tensor = None, cin_b: int = 0,
              res: Tensor = None, act: int = F_.ACT_RELU, alpha: float = 1.0, out_nchw_f32: bool = False,
              aux_mode: int = F_.AUX_NONE):
        wp = self._pack(name, conv, cin_a, x.shape[-1], cin_b, 0 if xb is None else xb.shape[-1])
        return F_.conv3x3(x, wp, conv.bias.detach(), conv.out_channels, src_b=xb, res=res, act=act, alpha=alpha,
                          out_nchw_f32=out_nchw_f32, aux_mode=aux_mode)

    def _pair(self, name: str, conv_a: nn.Conv2d, conv_b: nn.Conv2d, x: Tensor, act2: int, alpha: float,
              keep_mid: bool = False, emit_bits: bool = False):
        """alpha * act2(conv_b(relu(conv_a(x)))) + x as one launch (the intermediate map stays in LDS);
        SRHIP_NO_PAIR=1 in the environment runs the two single-layer launches instead (A/B timing).
        ``emit_bits`` (training): returns (out, mid, bits) with the ReLU pattern of ``mid`` in the form
        ``_pair_backward`` reads, or bits = None where the one-launch form does not apply."""
        ch = x.shape[-1]
        if self.use_pair_kernel and ch == 64 and conv_a.out_channels == 64 and conv_b.out_channels == 64:
            wa, wb = self._pack(name + "a", conv_a, ch, ch), self._pack(name + "b", conv_b, ch, ch)
            if emit_bits and act2 == F_.ACT_NONE and os.environ.get("SRHIP_NO_PAIR_BWD", "0") != "1":
                return F_.conv3x3_pair(x, wa, conv_a.bias.detach(), wb, conv_b.bias.detach(), act2=act2, alpha=alpha,
                                       emit_bits=True)
            res = F_.conv3x3_pair(x, wa, conv_a.bias.detach(), wb, conv_b.bias.detach(), act2=act2, alpha=alpha,
                                  keep_mid=keep_mid or emit_bits)
            return res + (None,) if emit_bits else res
        t = self._conv(name + "a", conv_a, x, ch)
        out = self._conv(name + "b", conv_b, t, ch, res=x, act=act2, alpha=alpha)
        if emit_bits:
            return out, t, None
        return (out, t) if keep_mid else out

    def _pair_backward(self, name: str, conv_a: nn.Conv2d, conv_b: nn.Conv2d, dz: Tensor, bits: Tensor, alpha: float):
        """Both data gradients of the block out = x + alpha * conv_b(relu(conv_a(x))) as one launch: from dz = dL/d(out)
        and the ReLU pattern the forward launch emitted, returns (dL/dx, g) with g = conv_b^T(dz) * [mid > 0], the
        gradient of conv_a's output WITHOUT the factor alpha (the caller scales the weight gradient formed from it)."""
        ch = dz.shape[-1]
        packs = []
        for tag, conv in (("b", conv_b), ("a", conv_a)):
            w = conv.weight
            packs.append(self._cached_pack(f"{name}{tag}^T0", w, lambda w=w: F_.conv3x3_pack_weights(
                w.detach(), conv.out_channels, ch, cout=conv.in_channels, cout_off=0, transpose_flip=True)))
        return F_.conv3x3_pair(dz, packs[0], None, packs[1], None, act2=F_.ACT_NONE, alpha=alpha, apply_bits=bits)

    def _pair_head(self, name: str, conv_a: nn.Conv2d, conv_b: nn.Conv2d, x: Tensor, keep_mid: bool = False):
        """conv_b(relu(conv_a(x))) as fp32 NCHW, the network's last two layers (model.py:111-115), as one launch when
        conv_a is 64 -> 64 and conv_b has at most 32 outputs; the 64-channel map between them stays in LDS."""
        ch = x.shape[-1]
        if self.use_pair_kernel and os.environ.get("SRHIP_NO_PAIR_HEAD", "0") != "1" and ch == 64 and conv_a.out_channels == 64 \
                and conv_b.out_channels <= 32:
            wa, wb = self._pack(name + "0", conv_a, ch, ch), self._pack(name + "2", conv_b, ch, ch)
            return F_.conv3x3_pair_head(x, wa, conv_a.bias.detach(), wb, conv_b.bias.detach(), conv_b.out_channels,
                                        keep_mid=keep_mid)
        t = self._conv(name + "0", conv_a, x, ch)
        out = self._conv(name + "2", conv_b, t, ch, act=F_.ACT_NONE, out_nchw_f32=True)
        return (out, t) if keep_mid else out

    # ------------------------------------------------------------------ forward
    def forward(self, batch: Dict[str, Union[Tensor, int]]):
        lr_frames = batch.pop("lr_frames")
        hr_buffers = batch.pop("hr_buffers")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .training import supersampler_forward_train  # autograd path (fp32 master weights, bf16 compute)
            batch["sr_frames"] = supersampler_forward_train(self, lr_frames[:, -1], hr_buffers[:, -1])
            return batch
        batch["sr_frames"] = self.infer(lr_frames[:, -1], hr_buffers[:, -1])
        return batch

    @torch.no_grad()
    def infer(self, lr: Tensor, hb: Tensor) -> Tensor:
        """lr (B,3,sh,sw) fp32, hb (B,10,th,tw) fp32 -> (B,3,th,tw) fp32.  hb may also be the HR trunk's input already in
        its device layout, (B,th,tw,16) bf16 NHWC (what ``functional.warp_concat_nhwc`` builds)."""
        prepared = hb.dtype == torch.bfloat16
        th, tw = hb.shape[1:3] if prepared else hb.shape[-2:]
        fc, bc, fr, bf = self.frame_channels, self.buffer_channels, self.FR_CH, self.BF_CH

        # frame_extractor @ LR (model.py:50-58,161)
        f = F_.nchw_f32_to_nhwc_bf16(lr.float())
        cin = fc
        for i, blk in enumerate(self.frame_extractor.blks):
            f = self._conv(f"fe{i}", blk[0], f, cin)
            cin = fr
        # fixed-ratio bilinear up-sampling with the reference's zero-padded border (model.py:153,162)
        up = F_.upsample_grid_bilinear_nhwc(f, th, tw)

        # gbuffers_extractor @ HR (model.py:61-79,167)
        g = hb if prepared else F_.nchw_f32_to_nhwc_bf16(hb.float())
        cin = bc
        blks = self.gbuffers_extractor.blks
        for i in range(0, len(blks), 2):
            g = self._conv(f"ge{i}", blks[i][0], g, cin)
            g = self._pair(f"ge{i + 1}", blks[i + 1][0][0], blks[i + 1][1][0], g, F_.ACT_RELU, 1.0)  # x + relu(conv(relu(conv(x))))
            cin = bf

        # main: concat(up, g) -> conv_in -> 6 x {x + 0.2 * conv(relu(conv(x)))}  (model.py:82-95,169)
        x = self._conv("main_in", self.main.conv_in[0], up, fr, xb=g, cin_b=bf)
        for i, blk in enumerate(self.main.conv_main):
            x = self._pair(f"m{i}", blk.blks[0], blk.blks[2], x, F_.ACT_NONE, 0.2)
        # after_main on concat(up, x) (model.py:170), final_conv (model.py:111-115,172)
        x = self._conv("after", self.after_main[0], up, fr, xb=x, cin_b=self.SR_CH)
        return self._pair_head("fin", self.final_conv[0], self.final_conv[2], x)
