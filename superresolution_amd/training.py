"""Training path of the default network: forward with saved activations and a hand-written
backward, both on libsrhip.so, exposed to torch as one ``autograd.Function``.

What the reference does with ``loss.backward()`` (core/trainer.py:160) through ATen's per-operator
autograd is done here layer by layer on NHWC bf16 tensors:

  * data gradient of a 3x3 layer = the forward MFMA kernel on ``transpose_flip`` weights, with the
    ReLU backward of the *producing* layer fused as a mask epilogue and the residual fan-in fused
    as an add epilogue;
  * weight and bias gradients = the MFMA weight-gradient kernel (fp32 accumulate, fp32 OIHW out);
  * the 0.2 of ``SRResidualBlock`` (core/models/model.py:33) travels as ``alpha`` / ``scale``;
  * the up-sampling backward is a gather with the frame extractor's last ReLU fused.

Master weights, gradients handed to the optimiser and the network output are fp32, activations and
activation gradients bf16 (fp32 accumulation inside every kernel).
"""
import os
from typing import List

import torch
from torch import Tensor

from . import functional as F_


def _wgrad(x: Tensor, dz: Tensor, conv, grads: dict, ci_real: int, ci_off: int = 0, scale: float = 1.0,
           with_bias: bool = True, red=None, last: bool = True) -> None:
    """Weight (+ bias) gradient of one 3x3 layer.  Without a reducer the gradients are new tensors handed back to
    autograd; with one (superresolution_amd.dp.FlatGradAllReducer.attach) they are accumulated straight into the
    parameter's view of the flat gradient buffer, and after the layer's ``last`` call the reducer is told, so that it
    can start the all-reduce of a completed bucket while the rest of the backward pass is still being queued."""
    wview = red.grad_view(conv.weight) if red is not None else None
    if wview is not None:
        bview = red.grad_view(conv.bias) if (with_bias and conv.bias is not None) else None
        F_.conv3x3_wgrad(x, dz, wview, bview, ci_real, conv.out_channels, ci_off=ci_off, accumulate=True, scale=scale)
        if last:
            red.mark_ready(conv.weight)
            if conv.bias is not None:
                red.mark_ready(conv.bias)
        return
    dw = grads.get(id(conv.weight))
    if dw is None:
        dw = torch.empty_like(conv.weight, memory_format=torch.contiguous_format)
        grads[id(conv.weight)] = dw
    db = None
    if with_bias and conv.bias is not None:
        db = torch.empty_like(conv.bias)
        grads[id(conv.bias)] = db
    F_.conv3x3_wgrad(x, dz, dw, db, ci_real, conv.out_channels, ci_off=ci_off, scale=scale)


class _SuperSamplerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, lr: Tensor, hb: Tensor, *params: Tensor) -> Tensor:
        m = model
        prepared = hb.dtype == torch.bfloat16   # the HR trunk's input already as (B,th,tw,16) bf16 NHWC
        th, tw = hb.shape[1:3] if prepared else hb.shape[-2:]
        fc, bc, fr, bf, sr = m.frame_channels, m.buffer_channels, m.FR_CH, m.BF_CH, m.SR_CH
        sv = {}
        # frame extractor @ LR (core/models/model.py:50-58,161)
        f = [F_.nchw_f32_to_nhwc_bf16(lr.float())]
        cin = fc
        for i, blk in enumerate(m.frame_extractor.blks):
            f.append(m._conv(f"fe{i}", blk[0], f[-1], cin))
            cin = fr
        up = F_.upsample_grid_bilinear_nhwc(f[-1], th, tw)
        # G-buffer extractor @ HR (model.py:61-79,167).  The backward needs the sign patterns of x = relu(conv0) and
        # of the branch output relu(conv_b) on their own: with 64-channel layers the launches that compute them also
        # write them out as 1 bit per element, and the residual sum x + relu(conv_b) is the second launch's epilogue;
        # other widths keep the branch output as a map of its own and add in a separate pass.
        blks = m.gbuffers_extractor.blks
        gb = []
        bits = bf == 64 and os.environ.get("SRHIP_NO_SIGNBITS", "0") != "1"
        prev, cin = (hb if prepared else F_.nchw_f32_to_nhwc_bf16(hb.float())), bc
        for i in range(0, len(blks), 2):
            if bits:
                x, xbits = m._conv(f"ge{i}", blks[i][0], prev, cin, aux_mode=F_.AUX_EMIT)
                t1 = m._conv(f"ge{i + 1}a", blks[i + 1][0][0], x, bf)
                nxt, t2 = m._conv(f"ge{i + 1}b", blks[i + 1][1][0], t1, bf, res=x, aux_mode=F_.AUX_EMIT)   # t2: sign bits only
            else:
                x, xbits = m._conv(f"ge{i}", blks[i][0], prev, cin), None
                t1 = m._conv(f"ge{i + 1}a", blks[i + 1][0][0], x, bf)
                t2 = m._conv(f"ge{i + 1}b", blks[i + 1][1][0], t1, bf)
                nxt = F_.add_bf16(x, t2)
            gb.append((prev, x, xbits, t1, t2))
            prev, cin = nxt, bf
        g = prev
        # main trunk (model.py:82-95,169)
        xs = [m._conv("main_in", m.main.conv_in[0], up, fr, xb=g, cin_b=bf)]
        # the two-source launch cannot emit its ReLU pattern itself: one light pass (reads the map, writes 1/16 of it)
        # lets the backward mask inside the last data-gradient launch of the trunk
        x0bits = F_.signbits_from_map(xs[0]) if (bits and sr == 64) else None
        ts, tbits = [], []
        for i, blk in enumerate(m.main.conv_main):
            # one launch per block; the intermediate map is written out as well (the backward needs it), and from the
            # second block on its ReLU pattern too, in the form the block's one-launch backward reads (the first
            # block's backward folds conv_in's ReLU into its last launch instead and keeps two launches)
            if i >= 1:
                x_next, t, tb = m._pair(f"m{i}", blk.blks[0], blk.blks[2], xs[-1], F_.ACT_NONE, 0.2, emit_bits=True)
            else:
                (x_next, t), tb = m._pair(f"m{i}", blk.blks[0], blk.blks[2], xs[-1], F_.ACT_NONE, 0.2, keep_mid=True), None
            ts.append(t)
            tbits.append(tb)
            xs.append(x_next)
        a = m._conv("after", m.after_main[0], up, fr, xb=xs[-1], cin_b=sr)
        out, c0 = m._pair_head("fin", m.final_conv[0], m.final_conv[2], a, keep_mid=True)   # one launch, c0 kept for the backward
        sv.update(f=f, up=up, gb=gb, g=g, xs=xs, ts=ts, tbits=tbits, a=a, c0=c0, lr_hw=tuple(lr.shape[-2:]), bits=bits, x0bits=x0bits)
        ctx.model, ctx.sv = m, sv
        return out

    @staticmethod
    def backward(ctx, dsr: Tensor):
        m, sv = ctx.model, ctx.sv
        fc, bc, fr, bf, sr = m.frame_channels, m.buffer_channels, m.FR_CH, m.BF_CH, m.SR_CH
        grads = {}
        dgrad = m._dgrad
        red = getattr(m, "_grad_reducer", None)   # data-parallel training: gradients go straight into its flat buffer
        if red is not None and (red.gather or any(red.grad_view(p) is None or p.grad is not red.grad_view(p) for p in m.parameters())):
            red = None
        if red is not None:
            red.begin_backward()
        f, up, gb, g, xs, ts, a, c0 = sv["f"], sv["up"], sv["gb"], sv["g"], sv["xs"], sv["ts"], sv["a"], sv["c0"]

        # final_conv (model.py:111-115,172)
        fin0, fin2 = m.final_conv[0], m.final_conv[2]
        dz = F_.nchw_f32_to_nhwc_bf16(dsr.float().contiguous())
        _wgrad(c0, dz, fin2, grads, sr, red=red)
        d = dgrad("fin2", fin2, dz, mask=c0)
        _wgrad(a, d, fin0, grads, sr, red=red)
        d = dgrad("fin0", fin0, d, mask=a)
        # after_main on concat(up, x6) (model.py:170)
        aft = m.after_main[0]
        _wgrad(up, d, aft, grads, fr, ci_off=0, red=red, last=False)
        _wgrad(xs[-1], d, aft, grads, sr, ci_off=fr, with_bias=False, red=red)
        dup = dgrad("after", aft, d, lo=0, n=fr)
        dx = dgrad("after", aft, d, lo=fr, n=sr)
        del d
        # SRResidualBlocks, last to first: x_{i+1} = x_i + 0.2 * conv_b(relu(conv_a(x_i)))
        for i in reversed(range(len(m.main.conv_main))):
            blk = m.main.conv_main[i].blks
            _wgrad(ts[i], dx, blk[2], grads, sr, scale=0.2, red=red)
            if sv["tbits"][i] is not None:   # both data gradients in one launch; the map between them carries no 0.2
                dx, g2 = m._pair_backward(f"m{i}", blk[0], blk[2], dx, sv["tbits"][i], 0.2)
                _wgrad(xs[i], g2, blk[0], grads, sr, scale=0.2, red=red)
                continue
            g2 = dgrad(f"m{i}b", blk[2], dx, alpha=0.2, mask=ts[i])
            _wgrad(xs[i], g2, blk[0], grads, sr, red=red)
            if i == 0 and sv["x0bits"] is not None:   # ... and the ReLU of conv_in applied to the fan-in sum, same launch
                dx = dgrad(f"m{i}a", blk[0], g2, res=dx, signbits=sv["x0bits"], aux_mode=F_.AUX_MASKBITS)
            else:
                dx = dgrad(f"m{i}a", blk[0], g2, res=dx)
        # conv_in on concat(up, g) (model.py:85,169)
        cin_ = m.main.conv_in[0]
        d = dx if sv["x0bits"] is not None else F_.relu_mask(dx, xs[0], out=dx)
        _wgrad(up, d, cin_, grads, fr, ci_off=0, red=red, last=False)
        _wgrad(g, d, cin_, grads, bf, ci_off=fr, with_bias=False, red=red)
        dup = dgrad("main_in", cin_, d, lo=0, n=fr, res=dup)
        # G-buffer extractor, last block first: out = x + relu(conv_b(relu(conv_a(x)))), x = relu(conv(prev)).
        # dg = gradient of a block's output, dt2 = dg masked by the branch's ReLU; with sign bits both come out of the
        # launch that computes dg, and the mask of x is applied by the launch that adds dg to the branch's gradient.
        bits = sv["bits"]
        blks = m.gbuffers_extractor.blks
        if bits:
            dg, dt2 = dgrad("main_in", cin_, d, lo=fr, n=bf, signbits=gb[-1][4], aux_mode=F_.AUX_DUALBITS)
        else:
            dg = dgrad("main_in", cin_, d, lo=fr, n=bf)
        del d, dx
        for b in reversed(range(len(gb))):
            prev, x, xbits, t1, t2 = gb[b]
            conv0, conva, convb = blks[2 * b][0], blks[2 * b + 1][0][0], blks[2 * b + 1][1][0]
            if not bits:
                dt2 = F_.relu_mask(dg, t2)
            _wgrad(t1, dt2, convb, grads, bf, red=red)
            dt1 = dgrad(f"ge{2 * b + 1}b", convb, dt2, mask=t1)
            _wgrad(x, dt1, conva, grads, bf, red=red)
            if bits:
                dzb = dgrad(f"ge{2 * b + 1}a", conva, dt1, res=dg, signbits=xbits, aux_mode=F_.AUX_MASKBITS)
            else:
                dxb = dgrad(f"ge{2 * b + 1}a", conva, dt1, res=dg)
                dzb = F_.relu_mask(dxb, x, out=dxb)
            _wgrad(prev, dzb, conv0, grads, bc if b == 0 else bf, red=red)
            if b > 0:
                if bits:
                    dg, dt2 = dgrad(f"ge{2 * b}", conv0, dzb, signbits=gb[b - 1][4], aux_mode=F_.AUX_DUALBITS)
                else:
                    dg = dgrad(f"ge{2 * b}", conv0, dzb)
        # frame extractor @ LR through the up-sampling (model.py:161-162)
        sh, sw = sv["lr_hw"]
        fe = m.frame_extractor.blks
        d = F_.upsample_grid_bilinear_bwd_nhwc(dup, sh, sw, relu_out=f[3])
        _wgrad(f[2], d, fe[2][0], grads, fr, red=red)
        d = dgrad("fe2", fe[2][0], d, mask=f[2])
        _wgrad(f[1], d, fe[1][0], grads, fr, red=red)
        d = dgrad("fe1", fe[1][0], d, mask=f[1])
        _wgrad(f[0], d, fe[0][0], grads, fc, red=red)
        ctx.sv = None
        return (None, None, None) + tuple(grads.get(id(p)) for p in m.parameters())


def supersampler_forward_train(model, lr: Tensor, hb: Tensor) -> Tensor:
    """Differentiable forward: (B,3,sh,sw), (B,10,th,tw) fp32 -> (B,3,th,tw) fp32 with a grad_fn that
    fills ``.grad`` of every parameter of ``model``.  hb may also be the trunk input in its device layout,
    (B,th,tw,16) bf16 NHWC; the inputs receive no gradient either way (the reference's do not require one)."""
    params: List[Tensor] = list(model.parameters())
    return _SuperSamplerFn.apply(model, lr, hb, *params)
