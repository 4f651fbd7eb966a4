"""ctypes/numpy front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

All arrays are C-contiguous float32 in the reference's NCHW layout.  The
model-level functions at the bottom restate the reference forward passes
(core/models/model.py:147-175, core/arch/RepRFN.py:327-339) on top of the
per-operator C functions, taking a reference-format state dict
(name -> numpy array, OIHW weights).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int


def build():
    """Compile liboracle.so in place (gcc only; used by __graft_entry__.build())."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(int(n))


# ---------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, padding=0, groups=1, act=0, slope=0.0):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), _i(N), _i(Cin), _i(H), _i(W), _i(Cout),
                         _i(kh), _i(kw), _i(sh), _i(sw), _i(ph), _i(pw), _i(groups), _i(act),
                         ctypes.c_float(slope))
    return y


def conv2d_bwd_input(dy, w, in_shape, stride=1, padding=0, groups=1):
    dy, w = _c(dy), _c(w)
    N, Cin, H, W = in_shape
    Cout, _, kh, kw = w.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    dx = np.empty(in_shape, np.float32)
    lib().orc_conv2d_bwd_input(_p(dy), _p(w), _p(dx), _i(N), _i(Cin), _i(H), _i(W), _i(Cout),
                               _i(kh), _i(kw), _i(sh), _i(sw), _i(ph), _i(pw), _i(groups))
    return dx


def conv2d_bwd_weight(x, dy, w_shape, stride=1, padding=0, groups=1, with_bias=True):
    x, dy = _c(x), _c(dy)
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w_shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    dw = np.empty(w_shape, np.float32)
    db = np.empty((Cout,), np.float32) if with_bias else None
    lib().orc_conv2d_bwd_weight(_p(x), _p(dy), _p(dw), _p(db), _i(N), _i(Cin), _i(H), _i(W),
                                _i(Cout), _i(kh), _i(kw), _i(sh), _i(sw), _i(ph), _i(pw), _i(groups))
    return dw, db


def make_coord(h, w):
    out = np.empty((1, 2, h, w), np.float32)
    lib().orc_make_coord(_p(out), _i(h), _i(w))
    return out


_MODES = {"bilinear": 0, "nearest": 1}


def grid_sample(inp, grid, mode="bilinear", align_corners=False):
    inp, grid = _c(inp), _c(grid)
    N, C, H, W = inp.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_grid_sample(_p(inp), _p(grid), _p(out), _i(N), _i(C), _i(H), _i(W), _i(Ho), _i(Wo),
                          _i(_MODES[mode]), _i(int(align_corners)))
    return out


def grid_sample_bwd(inp, grid, dout, align_corners=False, need_dgrid=True):
    inp, grid, dout = _c(inp), _c(grid), _c(dout)
    N, C, H, W = inp.shape
    _, Ho, Wo, _ = grid.shape
    din = np.zeros_like(inp)
    dgrid = np.empty_like(grid) if need_dgrid else None
    lib().orc_grid_sample_bwd(_p(inp), _p(grid), _p(dout), _p(din), _p(dgrid), _i(N), _i(C), _i(H),
                              _i(W), _i(Ho), _i(Wo), _i(int(align_corners)))
    return din, dgrid


def upsample_grid_bilinear(inp, th, tw):
    inp = _c(inp)
    N, C, sh, sw = inp.shape
    out = np.empty((N, C, th, tw), np.float32)
    lib().orc_upsample_grid_bilinear(_p(inp), _p(out), _i(N), _i(C), _i(sh), _i(sw), _i(th), _i(tw))
    return out


def warp_grid(motion):
    motion = _c(motion)
    N, _, h, w = motion.shape
    grid = np.empty((N, h, w, 2), np.float32)
    lib().orc_warp_grid(_p(motion), _p(grid), _i(N), _i(h), _i(w))
    return grid


def backward_warp(frame, motion, mode="nearest", align_corners=True):
    frame, motion = _c(frame), _c(motion)
    N, C, h, w = frame.shape
    out = np.empty_like(frame)
    lib().orc_backward_warp(_p(frame), _p(motion), _p(out), _i(N), _i(C), _i(h), _i(w),
                            _i(_MODES[mode]), _i(int(align_corners)))
    return out


def rel_coord(th, tw, sh, sw):
    out = np.empty((1, 2, th, tw), np.float32)
    lib().orc_rel_coord(_p(out), _i(th), _i(tw), _i(sh), _i(sw))
    return out


def pixel_shuffle(x, r):
    x = _c(x)
    N, Crr, H, W = x.shape
    C = Crr // (r * r)
    out = np.empty((N, C, H * r, W * r), np.float32)
    lib().orc_pixel_shuffle(_p(x), _p(out), _i(N), _i(C), _i(H), _i(W), _i(r))
    return out


def pixel_unshuffle(x, r):
    x = _c(x)
    N, C, Hr, Wr = x.shape
    H, W = Hr // r, Wr // r
    out = np.empty((N, C * r * r, H, W), np.float32)
    lib().orc_pixel_unshuffle(_p(x), _p(out), _i(N), _i(C), _i(H), _i(W), _i(r))
    return out


def nearest_down(x, scale_factor):
    x = _c(x)
    N, C, H, W = x.shape
    Ho, Wo = int(np.floor(H * scale_factor)), int(np.floor(W * scale_factor))
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_nearest_down(_p(x), _p(out), _i(N * C), _i(H), _i(W), ctypes.c_double(scale_factor),
                           _i(Ho), _i(Wo))
    return out


def layernorm_cf(x, weight, bias, eps=1e-6):
    x, weight, bias = _c(x), _c(weight), _c(bias)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_layernorm_cf(_p(x), _p(weight), _p(bias), _p(y), _i(N), _i(C), _i(H * W),
                           ctypes.c_float(eps))
    return y


def layernorm_cf_bwd(x, weight, dy, eps=1e-6):
    """-> (dx, dweight, dbias) of the channels-first LayerNorm."""
    x, weight, dy = _c(x), _c(weight), _c(dy)
    N, C, H, W = x.shape
    dx, dw, db = np.empty_like(x), np.empty_like(weight), np.empty_like(weight)
    lib().orc_layernorm_cf_bwd(_p(x), _p(weight), _p(dy), _p(dx), _p(dw), _p(db), _i(N), _i(C), _i(H * W),
                               ctypes.c_float(eps))
    return dx, dw, db


def disocclusion(cur_frame, cur_buffer, hist_frames, hist_buffer):
    """-> (mask (B,F,1,h,w) bool, marked_frames (B*F,3,h,w), aux (B*F,14,h,w), enhance_weight (B,1,h,w))."""
    cur_frame, cur_buffer, hist_frames, hist_buffer = _c(cur_frame), _c(cur_buffer), _c(hist_frames), _c(hist_buffer)
    B, Fh, _, H, W = hist_frames.shape
    mask = np.empty((B, Fh, 1, H, W), np.uint8)
    marked = np.empty((B * Fh, 3, H, W), np.float32)
    aux = np.empty((B * Fh, 14, H, W), np.float32)
    ew = np.empty((B, 1, H, W), np.float32)
    lib().orc_disocclusion(_p(cur_frame), _p(cur_buffer), _p(hist_frames), _p(hist_buffer),
                           mask.ctypes.data_as(ctypes.c_void_p), _p(marked), _p(aux), _p(ew), _i(B), _i(Fh), _i(H), _i(W))
    return mask.astype(bool), marked, aux, ew


def clamp_depth(depth, thres1=3.0, thres2=7.0):
    depth = _c(depth)
    out = np.empty_like(depth)
    lib().orc_clamp_depth(_p(depth), _p(out), ctypes.c_longlong(depth.size), ctypes.c_float(thres1), ctypes.c_float(thres2))
    return out


def dilate_velocity(motion, depth, ksize=3):
    """core/tools/math.py:80-100 per sample: motion (B,H,W,2), depth (B,1,H,W) -> (B,H,W,2)."""
    motion, depth = _c(motion), _c(depth)
    B, H, W, _ = motion.shape
    out = np.empty_like(motion)
    lib().orc_dilate_velocity(_p(motion), _p(depth), _p(out), _i(B), _i(H), _i(W), _i(ksize))
    return out


def load_and_augment(arrays, crop=None, hflip=False, vflip=False, velocity=False):
    """core/data/supersampling_sequence.py:153-200 on already-loaded `arr_0` arrays (fp16 CHW): crop (x1, y1, x2, y2),
    the two flips (hflip reverses axis 1, vflip axis 2), the motion-vector sign fix, cast to fp32, stack."""
    out = []
    for a in arrays:
        img = np.asarray(a)
        if crop is not None:
            x1, y1, x2, y2 = crop
            img = img[:, y1:y2, x1:x2]
        img = img.astype(np.float32)
        if hflip:
            img = img[:, ::-1, :].copy()
            if velocity:
                img[1] = -img[1]
        if vflip:
            img = img[:, :, ::-1].copy()
            if velocity:
                img[0] = -img[0]
        out.append(np.ascontiguousarray(img))
    return np.stack(out, axis=0)


def max_pool2d(x, k, s):
    x = _c(x)
    N, C, H, W = x.shape
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_max_pool2d(_p(x), _p(out), _i(N * C), _i(H), _i(W), _i(k), _i(s))
    return out


def resize_bilinear(x, Ho, Wo):
    x = _c(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_resize_bilinear(_p(x), _p(out), _i(N * C), _i(H), _i(W), _i(Ho), _i(Wo))
    return out


def conv_transpose2x2s2(x, w, b=None, relu=False):
    """nn.ConvTranspose2d(Cin, Cout, 2, 2) (+ReLU), w (Cin, Cout, 2, 2)."""
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    N, Cin, H, W = x.shape
    Cout = w.shape[1]
    out = np.empty((N, Cout, 2 * H, 2 * W), np.float32)
    lib().orc_conv_transpose2x2s2(_p(x), _p(w), _p(b), _p(out), _i(N), _i(Cin), _i(H), _i(W), _i(Cout), _i(int(relu)))
    return out


def resize_bilinear_align_corners(x, Ho, Wo):
    x = _c(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_resize_bilinear_ac(_p(x), _p(out), _i(N * C), _i(H), _i(W), _i(Ho), _i(Wo))
    return out


def gradient_map(x, eps=1e-6):
    x = _c(x)
    N, C, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_gradient_map(_p(x), _p(out), _i(N * C), _i(H), _i(W), ctypes.c_float(eps))
    return out


# ---------------------------------------------------------------------------
# model-level restatements
# ---------------------------------------------------------------------------
def supersampler_forward(sd, lr_frames, hr_buffers, return_intermediates=False):
    """core/models/model.py:147-175 on a reference state dict `sd`.

    lr_frames (B,L,3,sh,sw), hr_buffers (B,L,10,th,tw) -> sr_frames (B,3,th,tw).
    """
    def ca(x, key, act=1):  # ConvActivation, model.py:10-14
        return conv2d(x, sd[key + ".weight"], sd[key + ".bias"], 1, 1, act=act)

    lr = _c(lr_frames[:, -1])
    hb = _c(hr_buffers[:, -1])
    th, tw = hb.shape[-2:]
    # frame_extractor, model.py:50-58 (require_first=False)
    f = lr
    for i in range(3):
        f = ca(f, f"frame_extractor.blks.{i}.0")
    fmap_curr = f
    up = upsample_grid_bilinear(fmap_curr, th, tw)  # model.py:153,162
    # gbuffers_extractor, model.py:61-79
    g = hb
    for i in range(3):
        g = ca(g, f"gbuffers_extractor.blks.{2 * i}.0")
        r = ca(g, f"gbuffers_extractor.blks.{2 * i + 1}.0.0")
        r = ca(r, f"gbuffers_extractor.blks.{2 * i + 1}.1.0")
        g = g + r  # Residual, model.py:17-19
    fmap_buff = g
    # main, model.py:82-95
    x = ca(np.concatenate((up, fmap_buff), axis=1), "main.conv_in.0")
    for i in range(6):
        t = ca(x, f"main.conv_main.{i}.blks.0")
        t = ca(t, f"main.conv_main.{i}.blks.2", act=0)
        x = np.float32(0.2) * t + x  # model.py:33
    x = ca(np.concatenate((up, x), axis=1), "after_main.0")
    x = ca(x, "final_conv.0")
    sr = ca(x, "final_conv.2", act=0)
    if return_intermediates:
        return sr, {"fmap_curr": fmap_curr, "up": up, "fmap_buff": fmap_buff}
    return sr


def recurrent_forward(sd, lr_frames, hr_buffers, prev_sr=None, velocity=None):
    """Composition the recurrent variant is held to: the previous output warped by the motion vectors
    (core/tools/math.py:10-17 with the arguments of core/models/model_final.py:18,30: bilinear, align_corners=True)
    concatenated behind the G-buffers, then core/models/model.py:147-175 on the wider buffer tensor.
    hr_buffers (B,L,10,th,tw), prev_sr (B,3,th,tw) or None (zeros), velocity (B,2,th,tw)."""
    hb = _c(hr_buffers[:, -1])
    if prev_sr is None:
        cf = sd["gbuffers_extractor.blks.0.0.weight"].shape[1] - hb.shape[1]
        warped = np.zeros((hb.shape[0], cf) + hb.shape[2:], np.float32)
    else:
        warped = backward_warp(_c(prev_sr), _c(velocity), "bilinear", True)
    wide = np.concatenate((hb, warped), axis=1)[:, None]
    return supersampler_forward(sd, lr_frames, wide)


def bf16_round(a):
    """Round an fp32 array to the nearest bf16 value (ties to even), returned as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(np.shape(a))


def supersampler_forward_backward(sd, lr_frames, hr_buffers, dsr_fn, emulate_bf16=False, trace=None, pair_backward=True):
    """Forward (core/models/model.py:147-175) and the backward ATen's autograd performs for it
    (loss.backward(), core/trainer.py:160): returns (sr, grads) with grads keyed like the state
    dict.  `dsr_fn(sr)` supplies dL/d(sr) (the loss itself is outside this restatement).

    The backward is written in the order the GPU path evaluates it (ReLU masks folded into the product that a
    data-gradient launch writes, residual fan-in added there too), which in fp32 is the same arithmetic as autograd's
    up to summation order; this mode is pinned to the reference's gradients.  ``emulate_bf16=True`` additionally rounds
    to bf16 every tensor the GPU path STORES as bf16 -- inputs, packed weights, every activation and every activation
    gradient, once per kernel output -- while accumulation, the network output and all parameter gradients stay fp32:
    the GPU gradients are held to that mode at a tight tolerance."""
    r = bf16_round if emulate_bf16 else (lambda a: a)
    relu = lambda a: np.maximum(a, np.float32(0))
    lr = r(_c(lr_frames[:, -1]))
    hb = r(_c(hr_buffers[:, -1]))
    th, tw = hb.shape[-2:]
    grads = {}
    W = {k: (r(v) if k.endswith(".weight") else v) for k, v in sd.items()}     # weights are rounded when packed

    def conv(x, key):          # pre-activation, fp32
        return conv2d(x, W[key + ".weight"], W[key + ".bias"], 1, 1, act=0)

    def wgrad(x, dz, key, scale=None):
        dw, db = conv2d_bwd_weight(x, dz, W[key + ".weight"].shape, padding=1)
        if scale is not None:
            dw, db = np.float32(scale) * dw, np.float32(scale) * db
        grads[key + ".weight"], grads[key + ".bias"] = dw, db

    def dgrad(dz, key, x_shape):
        return conv2d_bwd_input(dz, W[key + ".weight"], x_shape, padding=1)

    # ---- forward, keeping every activation (each one kernel output, rounded once)
    f = [lr]
    for i in range(3):
        f.append(r(relu(conv(f[-1], f"frame_extractor.blks.{i}.0"))))
    up = r(upsample_grid_bilinear(f[-1], th, tw))
    gb, g = [], hb
    for i in range(3):
        x = r(relu(conv(g, f"gbuffers_extractor.blks.{2 * i}.0")))
        t1 = r(relu(conv(x, f"gbuffers_extractor.blks.{2 * i + 1}.0.0")))
        t2 = relu(conv(t1, f"gbuffers_extractor.blks.{2 * i + 1}.1.0"))     # never stored: the layer's epilogue adds x and
        gb.append((g, x, t1, t2))                                            # keeps only the sign pattern of t2
        g = r(x + t2)
    cat0 = np.concatenate((up, g), axis=1)
    xs = [r(relu(conv(cat0, "main.conv_in.0")))]
    ts = []
    for i in range(6):
        ts.append(r(relu(conv(xs[-1], f"main.conv_main.{i}.blks.0"))))
        xs.append(r(np.float32(0.2) * conv(ts[-1], f"main.conv_main.{i}.blks.2") + xs[-1]))   # model.py:33, one epilogue
    cat1 = np.concatenate((up, xs[-1]), axis=1)
    a = r(relu(conv(cat1, "after_main.0")))
    c0 = r(relu(conv(a, "final_conv.0")))
    sr = conv(c0, "final_conv.2")

    # ---- backward; dz_* are gradients of pre-activations
    nf = up.shape[1]
    dz = r(_c(dsr_fn(sr)))
    wgrad(c0, dz, "final_conv.2")
    dz = r(dgrad(dz, "final_conv.2", c0.shape) * (c0 > 0))
    wgrad(a, dz, "final_conv.0")
    dz = r(dgrad(dz, "final_conv.0", a.shape) * (a > 0))
    wgrad(cat1, dz, "after_main.0")
    d = dgrad(dz, "after_main.0", cat1.shape)
    dup, dx = r(d[:, :nf]), r(d[:, nf:])
    for i in reversed(range(6)):
        kb, ka = f"main.conv_main.{i}.blks.2", f"main.conv_main.{i}.blks.0"
        wgrad(ts[i], dx, kb, scale=0.2)
        if i >= 1 and pair_backward:
            # one launch carries both data gradients: the map between them is stored UNSCALED and the block's 0.2
            # multiplies the fp32 results behind it (weight gradient, data gradient) instead
            g2 = r(dgrad(dx, kb, ts[i].shape) * (ts[i] > 0))
            wgrad(xs[i], g2, ka, scale=0.2)
            dx = r(np.float32(0.2) * dgrad(g2, ka, xs[i].shape) + dx)
        else:
            g2 = r(np.float32(0.2) * dgrad(dx, kb, ts[i].shape) * (ts[i] > 0))
            wgrad(xs[i], g2, ka)
            dx = r(dgrad(g2, ka, xs[i].shape) + dx)
    dz = dx * (xs[0] > 0)
    wgrad(cat0, dz, "main.conv_in.0")
    d = dgrad(dz, "main.conv_in.0", cat0.shape)
    dup = r(d[:, :nf] + dup)
    dg = r(d[:, nf:])
    for i in reversed(range(3)):
        prev, x, t1, t2 = gb[i]
        k0, ka, kb = (f"gbuffers_extractor.blks.{2 * i}.0", f"gbuffers_extractor.blks.{2 * i + 1}.0.0",
                      f"gbuffers_extractor.blks.{2 * i + 1}.1.0")
        dt2 = dg * (t2 > 0)
        wgrad(t1, dt2, kb)
        dt1 = r(dgrad(dt2, kb, t1.shape) * (t1 > 0))
        wgrad(x, dt1, ka)
        dzb = r(dgrad(dt1, ka, x.shape) + dg) * (x > 0)
        wgrad(prev, dzb, k0)
        if i > 0:
            dg = r(dgrad(dzb, k0, prev.shape))
    grid = np.broadcast_to(np.transpose(make_coord(th, tw), (0, 2, 3, 1)), (lr.shape[0], th, tw, 2)).copy()
    d, _ = grid_sample_bwd(f[3], grid, dup, align_corners=False, need_dgrid=False)
    dz = r(d * (f[3] > 0))
    if trace is not None:      # intermediate maps for debugging a mismatch (NCHW)
        trace.update(up=up, g=g, x0=xs[0], x6=xs[-1], a=a, c0=c0, f3=f[3], dup=dup, dz_fe=dz)
    for i in reversed(range(3)):
        key = f"frame_extractor.blks.{i}.0"
        wgrad(f[i], dz, key)
        if i > 0:
            dz = r(dgrad(dz, key, f[i].shape) * (f[i] > 0))
    return sr, grads


def reprfn_forward_deploy(sd, x):
    """core/arch/RepRFN.py:327-339 in deploy form (state dict after switch_to_deploy())."""
    def conv(x, key, stride=1, padding=1, act=0, slope=0.0):
        return conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride, padding, act=act, slope=slope)

    def esa(x, p):  # RepRFN.py:295-308
        c1_ = conv(x, p + ".conv1", padding=0)
        c1 = conv(c1_, p + ".conv2", stride=2, padding=0)
        v_max = max_pool2d(c1, 7, 3)
        c3 = conv(v_max, p + ".conv3", padding=1)
        c3 = resize_bilinear(c3, x.shape[2], x.shape[3])
        cf = conv(c1_, p + ".conv_f", padding=0)
        c4 = conv(c3 + cf, p + ".conv4", padding=0)
        m = np.float32(1.0) / (np.float32(1.0) + np.exp(-c4, dtype=np.float32))
        return x * m

    x = _c(x)
    feat = conv(x, "fea_conv")
    out = feat
    for k in range(1, 5):
        p = f"reprfb{k}"
        inp = out
        o = inp
        for j in range(1, 4):
            o = conv(o, f"{p}.repblock{j}.rbr_reparam", act=2, slope=0.05)
        o = conv(o, f"{p}.conv3", act=2, slope=0.05)
        o = inp + o
        out = esa(o, f"{p}.esa")
    out = conv(out, "lr_conv")
    out = out + feat
    out = conv(out, "upsampler.conv")
    r = int(round((out.shape[1] // 3) ** 0.5))
    return pixel_shuffle(out, r)
