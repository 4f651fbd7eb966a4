"""Tensor-level wrappers over the C ABI of libsrhip.so.

Every function takes/returns torch tensors that live on a ROCm device; torch is used only to
own device memory and to name the current HIP stream.  NHWC tensors are plain contiguous
``(N, H, W, C)`` bf16 tensors (not torch "channels_last" views).
"""
from typing import Optional

import ctypes
import os

import torch

from ._lib import Conv1x1ChainArgs, ConvGemmArgs, Conv3x3Args, Conv3x3PairArgs, check, lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
RES_ADD, RES_MASK, RES_PRE = 0, 1, 2
AUX_NONE, AUX_EMIT, AUX_MASKBITS, AUX_DUALBITS = 0, 1, 2, 3
_MODES = {"bilinear": 0, "nearest": 1}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def invalidate_pack_cache() -> None:
    """Drop every cached packed-weight copy made by conv2d_auto / conv_transpose2x2s2_auto.  The cache is keyed on each
    weight tensor's version counter, which in-place writes through ``Parameter.data`` do not bump."""
    _pack_cache.clear()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a tensor on the GPU (libsrhip has no CPU path)")
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on the CURRENT device's stream: a tensor elsewhere would be read from the wrong GPU
        raise ValueError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                         "(wrap the call in torch.cuda.device(tensor.device))")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


# ------------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------------
def nchw_f32_to_nhwc_bf16(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    x = _req(x, torch.float32, "x")
    n, c, h, w = x.shape
    cpad = pad16(c) if cpad is None else cpad
    out = torch.empty((n, h, w, cpad), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), n, c, h, w, cpad, _stream()))
    return out


def nhwc_bf16_to_nchw_f32(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    x = _req(x, torch.bfloat16, "x")
    n, h, w, cs = x.shape
    c = cs if c is None else c
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib.srhip_nhwc_bf16_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, cs, _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# 3x3 MFMA convolution
# ------------------------------------------------------------------------------------------------
def conv3x3_pack_weights(weight: torch.Tensor, cin_a_real: int, ca_pad: int, cin_b_real: int = 0, cb_pad: int = 0,
                         cout: Optional[int] = None, cout_off: int = 0, transpose_flip: bool = False) -> torch.Tensor:
    """fp32 (O,I,3,3) master weights -> packed bf16 fragments (opaque uint8 tensor)."""
    weight = _req(weight, torch.float32, "weight")
    w_o, w_i, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError("conv3x3_pack_weights: kernel must be 3x3")
    if cout is None:
        cout = w_i if transpose_flip else w_o
    nbytes = lib.srhip_conv3x3_packed_bytes(ca_pad + cb_pad, cout)
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    check(lib.srhip_conv3x3_pack_weights(weight.data_ptr(), packed.data_ptr(), w_o, w_i, cout, cout_off, cin_a_real,
                                         ca_pad, cin_b_real, cb_pad, int(transpose_flip), _stream()))
    return packed


def conv3x3(src_a: torch.Tensor, wpack: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
            src_b: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, act: int = ACT_NONE,
            slope: float = 0.0, alpha: float = 1.0, out_nchw_f32: bool = False,
            cout_stride: Optional[int] = None, out: Optional[torch.Tensor] = None,
            res_mode: int = RES_ADD, ca_slice: Optional[tuple] = None, out_off: Optional[int] = None,
            signbits: Optional[torch.Tensor] = None, aux_mode: int = AUX_NONE, out2: Optional[torch.Tensor] = None):
    """dst = alpha * act(conv3x3(concat(src_a, src_b)) + bias) + res   (NHWC bf16 activations);
    with ``res_mode=RES_MASK``: dst = alpha * act(...) * (res > 0 ? 1 : m), m = 0 or (act NONE) m = ``slope``;
    with ``res_mode=RES_PRE``: dst = act(alpha * (conv + bias) + res) (partial sums of layers wider than one launch).

    ``ca_slice=(off, n)``: read only channels [off, off+n) of src_a (n a multiple of 16, off of 8) - no copy.
    ``out_off=k`` (with ``out`` a wider NHWC tensor): write pad16(cout) channels at channel offset k of ``out``;
    ``res`` (if given) is then a tensor of out's shape and is read at the same offset (it may be ``out`` itself).

    Sign-bit forms (64-channel layers; ``signbits`` from :func:`signbits_empty`), with X = alpha * act(conv + bias):
    ``AUX_EMIT`` also writes the bits of X > 0 and returns (dst, signbits); ``AUX_MASKBITS`` returns (X + res) where the bit
    is set and 0 elsewhere; ``AUX_DUALBITS`` returns (X, X where the bit is set and 0 elsewhere)."""
    src_a = _req(src_a, torch.bfloat16, "src_a")
    n, h, w, ca_full = src_a.shape
    a_ptr, ca, ca_stride = src_a.data_ptr(), ca_full, 0
    if ca_slice is not None:
        off, ca = ca_slice
        if src_b is not None or off % 8 or ca % 16 or off < 0 or off + ca > ca_full:
            raise ValueError("conv3x3: bad ca_slice (single source; offset multiple of 8, width multiple of 16, inside the tensor)")
        a_ptr, ca_stride = a_ptr + 2 * off, ca_full
    cb = 0
    if src_b is not None:
        src_b = _req(src_b, torch.bfloat16, "src_b")
        if src_b.shape[:3] != src_a.shape[:3]:
            raise ValueError("conv3x3: src_a and src_b disagree on (N,H,W)")
        cb = src_b.shape[3]
    if bias is not None:
        bias = _req(bias, torch.float32, "bias")
    cout_w, d_off = 0, 0
    if out_off is not None:
        if out is None or out_nchw_f32 or out.dtype != torch.bfloat16 or not out.is_contiguous() or tuple(out.shape[:3]) != (n, h, w):
            raise ValueError("conv3x3: out_off needs a contiguous NHWC bf16 `out` of the same (N,H,W)")
        cout_stride, cout_w, d_off = out.shape[3], pad16(cout), 2 * out_off
        if out_off % 8 or out_off < 0 or out_off + cout_w > cout_stride:
            raise ValueError("conv3x3: the written channel slice must lie inside `out` at an offset that is a multiple of 8")
    if cout_stride is None:
        cout_stride = pad16(cout)
    if out_nchw_f32:
        if out is None:
            out = torch.empty((n, cout, h, w), dtype=torch.float32, device=src_a.device)
    else:
        if out is None:
            out = torch.empty((n, h, w, cout_stride), dtype=torch.bfloat16, device=src_a.device)
        if res is not None:
            res = _req(res, torch.bfloat16, "res")
            if tuple(res.shape) != (n, h, w, cout_stride):
                raise ValueError("conv3x3: residual shape must equal the output shape")
    a = Conv3x3Args(src_a=a_ptr, src_b=_ptr(src_b), wpack=wpack.data_ptr(), bias=_ptr(bias),
                    res=None if res is None else res.data_ptr() + d_off, dst=out.data_ptr() + d_off, n=n, h=h, w=w, ca=ca, cb=cb,
                    cout=cout, cout_stride=cout_stride, act=act, slope=slope, alpha=alpha, dst_nchw_f32=int(out_nchw_f32),
                    res_mode=res_mode, ca_stride=ca_stride, cout_w=cout_w)
    if aux_mode != AUX_NONE:
        if aux_mode == AUX_EMIT and signbits is None:
            signbits = signbits_empty(n, h, w, src_a.device)
        if signbits is None or signbits.dtype != torch.int32 or not signbits.is_contiguous() or signbits.device != src_a.device \
                or tuple(signbits.shape) != (n, h, (w + 31) // 32, 64):
            raise ValueError("conv3x3: `signbits` must be the int32 tensor signbits_empty(n, h, w) gives")
        a.aux, a.aux_mode = signbits.data_ptr(), aux_mode
        if aux_mode == AUX_DUALBITS:
            if out2 is None:
                out2 = torch.empty_like(out)
            if out2.dtype != torch.bfloat16 or not out2.is_contiguous() or out2.shape != out.shape or out2.device != out.device:
                raise ValueError("conv3x3: `out2` must be a contiguous bf16 tensor of the output's shape")
            a.dst2 = out2.data_ptr()
    check(lib.srhip_conv3x3_fwd(a, _stream()))
    if aux_mode == AUX_EMIT:
        return out, signbits
    if aux_mode == AUX_DUALBITS:
        return out, out2
    return out


def signbits_empty(n: int, h: int, w: int, device) -> torch.Tensor:
    """Uninitialised sign-bit tensor of an (n, h, w, 64) map: 1 bit per element, (n, h, ceil(w/32), 64) int32."""
    t = torch.empty((n, h, (w + 31) // 32, 64), dtype=torch.int32, device=device)
    assert t.numel() * 4 == lib.srhip_conv3x3_signbits_bytes(n, h, w)
    return t


def signbits_from_map(y: torch.Tensor) -> torch.Tensor:
    """Sign bits (y > 0) of an (n, h, w, 64) bf16 map, in the layout the convolution's sign-bit forms consume."""
    y = _req(y, torch.bfloat16, "y")
    n, h, w, c = y.shape
    if c != 64:
        raise ValueError("signbits_from_map: 64-channel maps only")
    bits = signbits_empty(n, h, w, y.device)
    check(lib.srhip_signbits_from_map_bf16(y.data_ptr(), bits.data_ptr(), n, h, w, _stream()))
    return bits


def signbits_unpack(bits: torch.Tensor, w: int) -> torch.Tensor:
    """(n, h, ceil(w/32), 64) int32 sign bits -> (n, h, w, 64) bool (torch ops; for tests and debugging).
    Pixel x, channel c: word (x//32)*64 + (x%8)*8 + c//8, bit ((x%32)//8)*8 + c%8."""
    n, h, g, _ = bits.shape
    wd = bits.view(n, h, g, 8, 8)                              # [.., x%8, c//8]
    sh = torch.arange(32, device=bits.device, dtype=torch.int32).view(4, 8)  # [(x%32)//8, c%8]
    b = (wd[..., None, None] >> sh) & 1                        # n h g x8 c8 xq c%8
    b = b.permute(0, 1, 2, 5, 3, 4, 6).reshape(n, h, g * 32, 64)
    return b[:, :, :w].bool()


def conv3x3_pair(x: torch.Tensor, wpack1: torch.Tensor, bias1: Optional[torch.Tensor], wpack2: torch.Tensor,
                 bias2: Optional[torch.Tensor], act2: int = ACT_NONE, slope2: float = 0.0, alpha: float = 1.0,
                 keep_mid: bool = False, out: Optional[torch.Tensor] = None, emit_bits: bool = False,
                 apply_bits: Optional[torch.Tensor] = None):
    """alpha * act2(conv2(relu(conv1(x) + b1)) + b2) + x for a 64-channel NHWC bf16 map in ONE launch (the
    intermediate map stays in LDS).  ``keep_mid=True`` also returns relu(conv1(x) + b1) (the backward pass needs it).
    ``emit_bits=True`` (with keep_mid, act2 none): also returns the intermediate map's ReLU pattern as a bit tensor.
    ``apply_bits=bits``: the intermediate map is conv1(x) zeroed where the bit is clear instead of relu(conv1(x) + b1) --
    with transposed / flipped packs of a block's second and first layer this launch is the block's data gradient
    (returns (dL/d(in), masked gradient of the intermediate map))."""
    x = _req(x, torch.bfloat16, "x")
    if x.dim() != 4 or x.shape[-1] != 64:
        raise ValueError("conv3x3_pair: expected an (N,H,W,64) map")
    n, h, w, _ = x.shape
    if bias1 is not None:
        bias1 = _req(bias1, torch.float32, "bias1")
    if bias2 is not None:
        bias2 = _req(bias2, torch.float32, "bias2")
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != torch.bfloat16 or not out.is_contiguous() or out.data_ptr() == x.data_ptr():
        raise ValueError("conv3x3_pair: out must be a contiguous bf16 tensor of x's shape, distinct from x")
    bits, mode = None, 0
    if emit_bits or apply_bits is not None:
        if emit_bits and apply_bits is not None:
            raise ValueError("conv3x3_pair: emit_bits and apply_bits exclude each other")
        nbytes = lib.srhip_conv3x3_pair_bits_bytes(n, h, w)
        if emit_bits:
            bits, mode = torch.empty((nbytes,), dtype=torch.uint8, device=x.device), 1
        else:
            bits, mode = _req(apply_bits, torch.uint8, "apply_bits"), 2
            if bits.numel() != nbytes:
                raise ValueError("conv3x3_pair: apply_bits does not belong to a map of this shape")
        keep_mid = True
    mid = torch.empty_like(x) if keep_mid else None
    a = Conv3x3PairArgs(src=x.data_ptr(), wpack1=wpack1.data_ptr(), bias1=_ptr(bias1), wpack2=wpack2.data_ptr(), bias2=_ptr(bias2),
                        mid=_ptr(mid), dst=out.data_ptr(), n=n, h=h, w=w, act2=act2, slope2=slope2, alpha=alpha,
                        mid_bits=_ptr(bits), mid_bits_mode=mode)
    check(lib.srhip_conv3x3_pair_fwd(a, _stream()))
    if emit_bits:
        return out, mid, bits
    return (out, mid) if keep_mid else out


def conv3x3_pair_head(x: torch.Tensor, wpack1: torch.Tensor, bias1: Optional[torch.Tensor], wpack2: torch.Tensor,
                      bias2: Optional[torch.Tensor], cout2: int, act2: int = ACT_NONE, slope2: float = 0.0, alpha: float = 1.0,
                      keep_mid: bool = False):
    """alpha * act2(conv2(relu(conv1(x) + b1)) + b2) with a 64 -> cout2 <= 32 second layer, as fp32 NCHW (N, cout2, H, W), in
    ONE launch: the network's last two layers (core/models/model.py:111-115).  ``wpack2`` = conv3x3_pack_weights(w2, 64, 64)
    of the (cout2, 64, 3, 3) weight.  ``keep_mid=True`` also returns relu(conv1(x) + b1) as an NHWC bf16 map."""
    x = _req(x, torch.bfloat16, "x")
    if x.dim() != 4 or x.shape[-1] != 64:
        raise ValueError("conv3x3_pair_head: expected an (N,H,W,64) map")
    if not 1 <= cout2 <= 32:
        raise ValueError("conv3x3_pair_head: cout2 must be in 1..32")
    n, h, w, _ = x.shape
    if bias1 is not None:
        bias1 = _req(bias1, torch.float32, "bias1")
    if bias2 is not None:
        bias2 = _req(bias2, torch.float32, "bias2")
    out = torch.empty((n, cout2, h, w), dtype=torch.float32, device=x.device)
    mid = torch.empty_like(x) if keep_mid else None
    a = Conv3x3PairArgs(src=x.data_ptr(), wpack1=wpack1.data_ptr(), bias1=_ptr(bias1), wpack2=wpack2.data_ptr(), bias2=_ptr(bias2),
                        mid=_ptr(mid), dst=out.data_ptr(), n=n, h=h, w=w, act2=act2, slope2=slope2, alpha=alpha, cout2=cout2)
    check(lib.srhip_conv3x3_pair_fwd(a, _stream()))
    return (out, mid) if keep_mid else out


_wgrad_ws = {}


def _wgrad_workspace(device) -> torch.Tensor:
    """Partial-tile workspace of the weight-gradient kernel, one per (device, stream): launches on one stream are ordered
    and may share it; two streams running weight gradients at the same time must not (one partial tile set of 160 KB per CU: 42 MB each)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        with torch.cuda.device(device):
            ws = torch.empty((lib.srhip_conv3x3_wgrad_workspace_bytes(),), dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


def conv3x3_wgrad(x: torch.Tensor, dz: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor], ci_real: int,
                  co_real: int, ci_off: int = 0, accumulate: bool = False, scale: float = 1.0,
                  x_window=None, dz_window=None) -> None:
    """dw[:, ci_off:ci_off+ci_real] (=|+=) weight gradient of a 3x3 layer from its input x (N,H,W,ci_pad) and the
    gradient dz (N,H,W,co_pad) of its pre-activation output; db (=|+=) bias gradient (or None).
    x_window / dz_window = (first_channel, tile) read channels [first_channel, first_channel + tile) of a wider map in
    place (tile 16, 32 or 64; it may reach past the map's last channel, ci_real / co_real say how many channels count)."""
    x = _req(x, torch.bfloat16, "x")
    dz = _req(dz, torch.bfloat16, "dz")
    if x.shape[:3] != dz.shape[:3]:
        raise ValueError("conv3x3_wgrad: x and dz disagree on (N,H,W)")
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.dim() != 4 or tuple(dw.shape[2:]) != (3, 3):
        raise ValueError("conv3x3_wgrad: dw must be a contiguous fp32 (O,I,3,3) tensor")
    if dw.shape[0] < co_real:
        raise ValueError("conv3x3_wgrad: dw has fewer output channels than co_real")
    if db is not None and (db.dtype != torch.float32 or db.numel() < co_real):
        raise ValueError("conv3x3_wgrad: db must be fp32[co_real]")
    n, h, w, ci_pad = x.shape
    if x_window is None and dz_window is None:
        check(lib.srhip_conv3x3_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _ptr(db), _wgrad_workspace(x.device).data_ptr(),
                                      n, h, w, ci_pad, ci_real, ci_off, dw.shape[1], dz.shape[3], co_real, int(accumulate),
                                      scale, _stream()))
        return
    xo, xt = (0, ci_pad) if x_window is None else x_window
    zo, zt = (0, dz.shape[3]) if dz_window is None else dz_window
    check(lib.srhip_conv3x3_wgrad_windows(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _ptr(db), _wgrad_workspace(x.device).data_ptr(),
                                          n, h, w, xt, ci_real, ci_off, dw.shape[1], zt, co_real, int(accumulate), scale,
                                          ci_pad, xo, dz.shape[3], zo, _stream()))


WGRAD_MAX_JOBS = 32


def conv3x3_wgrad_multi(jobs, dw: torch.Tensor, db: Optional[torch.Tensor], ci_tile: int, co_tile: int,
                        accumulate: bool = False, scale: float = 1.0) -> None:
    """Several channel-slice pairs of ONE layer in one launch pair (srhip_conv3x3_wgrad_multi).  jobs: up to 32 tuples
    (x, x_first_channel, ci_real, ci_off, dz, dz_first_channel, co_real, co_off, with_bias): job j fills
    dw[co_off : co_off + co_real, ci_off : ci_off + ci_real] from channels [x_first_channel, + ci_tile) of the map x and
    [dz_first_channel, + co_tile) of the map dz (tiles 16, 32 or 64; a window may reach past its map's last channel), and,
    with_bias, db[co_off : co_off + co_real].  All maps share (N, H, W).  dw is the fp32 gradient of a (O,I,3,3) weight,
    or of a (O,I,1,1) weight (srhip_conv1x1_wgrad_multi: the centre tap alone)."""
    from ._lib import WgradJob
    if not 1 <= len(jobs) <= WGRAD_MAX_JOBS:
        raise ValueError("conv3x3_wgrad_multi: 1 .. %d jobs per call" % WGRAD_MAX_JOBS)
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.dim() != 4 or tuple(dw.shape[2:]) not in ((3, 3), (1, 1)):
        raise ValueError("conv3x3_wgrad_multi: dw must be a contiguous fp32 (O,I,3,3) or (O,I,1,1) tensor")
    taps = dw.shape[2] * dw.shape[3]
    if db is not None and (db.dtype != torch.float32 or not db.is_contiguous() or db.numel() < dw.shape[0]):
        raise ValueError("conv3x3_wgrad_multi: db must be fp32[O]")
    arr = (WgradJob * len(jobs))()
    shape = None
    w_i = dw.shape[1]
    for j, (x, xo, ci_real, ci_off, dz, zo, co_real, co_off, with_bias) in enumerate(jobs):
        x = _req(x, torch.bfloat16, "x")
        dz = _req(dz, torch.bfloat16, "dz")
        if shape is None:
            shape = tuple(x.shape[:3])
        if tuple(x.shape[:3]) != shape or tuple(dz.shape[:3]) != shape:
            raise ValueError("conv3x3_wgrad_multi: the maps disagree on (N,H,W)")
        if co_off < 0 or co_off + co_real > dw.shape[0]:
            raise ValueError("conv3x3_wgrad_multi: output-channel slice outside dw")
        a = arr[j]
        a.x, a.dz = x.data_ptr(), dz.data_ptr()
        a.dw = dw.data_ptr() + co_off * w_i * taps * 4
        a.db = (db.data_ptr() + co_off * 4) if (with_bias and db is not None) else None
        a.ci_real, a.ci_off, a.co_real = ci_real, ci_off, co_real
        a.x_cstride, a.x_coff, a.dz_cstride, a.dz_coff, a.reserved = x.shape[3], xo, dz.shape[3], zo, 0
    n, h, w = shape
    entry = lib.srhip_conv3x3_wgrad_multi if taps == 9 else lib.srhip_conv1x1_wgrad_multi
    check(entry(arr, len(jobs), _wgrad_workspace(jobs[0][0].device).data_ptr(), n, h, w, ci_tile, co_tile, w_i, int(accumulate),
                scale, _stream()))


def relu_mask(dy: torch.Tensor, y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy where y > 0 else 0 (bf16)."""
    dy = _req(dy, torch.bfloat16, "dy")
    y = _req(y, torch.bfloat16, "y")
    if dy.shape != y.shape:
        raise ValueError("relu_mask: shapes differ")
    out = torch.empty_like(dy) if out is None else out
    check(lib.srhip_relu_mask_bf16(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _stream()))
    return out


def add_bf16(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    a = _req(a, torch.bfloat16, "a")
    b = _req(b, torch.bfloat16, "b")
    if a.shape != b.shape:
        raise ValueError("add_bf16: shapes differ")
    out = torch.empty_like(a) if out is None else out
    check(lib.srhip_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


def scale_add_bf16(a: torch.Tensor, y: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """bf16(a + scale * y) with a 0-d fp32 device tensor as the scale (fmap + enhance_param * enhance,
    core/models/model_final.py:256-258, in the training graph): fp32 arithmetic, one rounding, one pass."""
    a, y = _req(a, torch.bfloat16, "a"), _req(y, torch.bfloat16, "y")
    scale = _req(scale.reshape(1), torch.float32, "scale")
    if a.shape != y.shape:
        raise ValueError("scale_add_bf16: shapes differ")
    out = torch.empty_like(a)
    check(lib.srhip_scale_add_bf16(a.data_ptr(), y.data_ptr(), scale.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


def scale_add_bwd_bf16(dout: torch.Tensor, y: torch.Tensor, scale: torch.Tensor):
    """Backward of scale_add_bf16 in one pass: returns (dy = bf16(scale * dout), dscale = sum(dout * y), 0-d fp32); the
    gradient of a is dout itself."""
    dout, y = _req(dout, torch.bfloat16, "dout"), _req(y, torch.bfloat16, "y")
    scale = _req(scale.reshape(1), torch.float32, "scale")
    if dout.shape != y.shape:
        raise ValueError("scale_add_bwd_bf16: shapes differ")
    dy = torch.empty_like(y)
    nparts = max(1, min(2048, (dout.numel() // 8 + 255) // 256))
    parts = torch.empty(nparts, dtype=torch.float32, device=y.device)
    check(lib.srhip_scale_add_bwd_bf16(dout.data_ptr(), y.data_ptr(), scale.data_ptr(), dy.data_ptr(), parts.data_ptr(), nparts,
                                       dout.numel(), _stream()))
    return dy, parts.sum()


def upsample_grid_bilinear_bwd_nhwc(dy: torch.Tensor, sh: int, sw: int, relu_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of upsample_grid_bilinear_nhwc: dy (N,th,tw,C) bf16 -> (N,sh,sw,C) bf16, optionally masked by the
    ReLU output that produced the source."""
    dy = _req(dy, torch.bfloat16, "dy")
    n, th, tw, c = dy.shape
    if relu_out is not None:
        relu_out = _req(relu_out, torch.bfloat16, "relu_out")
        if tuple(relu_out.shape) != (n, sh, sw, c):
            raise ValueError("upsample_grid_bilinear_bwd_nhwc: relu_out shape mismatch")
    out = torch.empty((n, sh, sw, c), dtype=torch.bfloat16, device=dy.device)
    check(lib.srhip_upsample_grid_bilinear_bwd_nhwc(dy.data_ptr(), _ptr(relu_out), out.data_ptr(), n, c, sh, sw, th, tw, _stream()))
    return out


def conv_transpose2x2s2(src: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], cin: int, act: int = ACT_NONE,
                        cout_stride: Optional[int] = None) -> torch.Tensor:
    """nn.ConvTranspose2d(cin, cout, 2, 2) (+ReLU) on NHWC bf16; weight is the module's fp32 (cin, cout, 2, 2) tensor."""
    src = _req(src, torch.bfloat16, "src")
    weight = _req(weight, torch.float32, "weight")
    n, h, w, cs = src.shape
    if weight.dim() != 4 or weight.shape[0] != cin or tuple(weight.shape[2:]) != (2, 2):
        raise ValueError("conv_transpose2x2s2: weight must be (cin, cout, 2, 2)")
    cout = weight.shape[1]
    cout_stride = pad16(cout) if cout_stride is None else cout_stride
    if bias is not None:
        bias = _req(bias, torch.float32, "bias")
    out = torch.empty((n, 2 * h, 2 * w, cout_stride), dtype=torch.bfloat16, device=src.device)
    check(lib.srhip_conv_transpose2x2s2(src.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), n, h, w, cin, cs, cout,
                                        cout_stride, act, _stream()))
    return out


def conv_transpose2x2s2_auto(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    """nn.ConvTranspose2d(cin, cout, 2, 2) (+activation) on the MFMA kernel: with stride = kernel size every output
    pixel (2y+i, 2x+j) depends on one input pixel only, out = W[:, :, i, j]^T x + b, i.e. a 1x1 convolution to
    4*cout channels (phase-major) followed by a depth-to-space copy.  `weight` (cin, cout, 2, 2) may be a Parameter
    (the rearranged weight and its packed form are cached).  Falls back to the direct kernel when cout % 8 != 0."""
    cin, cout = weight.shape[0], weight.shape[1]
    if cout % 8 != 0 or tuple(weight.shape[2:]) != (2, 2):
        return conv_transpose2x2s2(x, weight.detach(), None if bias is None else bias.detach(), cin, act=act)
    w1 = _cached_pack(weight, ("convT-as-1x1",),
                      lambda: weight.detach().float().permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1).contiguous())
    b1 = None if bias is None else _cached_pack(bias, ("convT-bias-x4",), lambda: bias.detach().float().repeat(4).contiguous())
    y = conv2d_auto(x, w1, b1, stride=1, padding=0, act=act)
    n, h, w, _ = x.shape
    out = torch.empty((n, 2 * h, 2 * w, cout), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_depth_to_space2_nhwc(y.data_ptr(), out.data_ptr(), n, h, w, cout, _stream()))
    return out


def gradient_map(x: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GradientMap (core/models/model_final.py:160-187) on an fp32 (N,C,H,W) tensor; `out` may be a contiguous
    slice of a larger tensor (every plane is processed on its own)."""
    x = _req(x, torch.float32, "x")
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    elif out.dtype != torch.float32 or out.shape != x.shape or not out.is_contiguous() or out.device != x.device:
        raise ValueError("gradient_map: out must be a contiguous fp32 tensor of x's shape on x's device")
    check(lib.srhip_gradient_map_f32(x.data_ptr(), out.data_ptr(), n * c, h, w, eps, _stream()))
    return out


_pack_cache = {}
# SRHIP_NO_CONV_GEMM=1 sends the layers outside the 3x3 kernel's native shape down the older path (64-channel 3x3 launches
# chained through bf16 partial sums); kept for A/B timing
use_conv_gemm = os.environ.get("SRHIP_NO_CONV_GEMM", "0") != "1"


def _cached_pack(weight: torch.Tensor, tag, build) -> torch.Tensor:
    """Packed weights of a long-lived weight tensor (a Parameter): rebuilt when the tensor object, its data
    or its version changes.  The entry dies with the tensor (weak reference), so a new tensor that happens to
    land on the same address can never hit a stale pack."""
    import weakref
    key = (id(weight), tag)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        return hit[2]
    packed = build()
    try:
        # (bound to the dict itself: at interpreter shutdown the module's globals are cleared before the last tensors die)
        ref = weakref.ref(weight, lambda _r, k=key, cache=_pack_cache: cache.pop(k, None))
    except TypeError:
        return packed
    _pack_cache[key] = (ref, (weight._version, weight.data_ptr()), packed)
    return packed


def _k_slices(cpad: int):
    """Input-channel slices of at most 64 channels (one MFMA source each)."""
    out, k0 = [], 0
    while k0 < cpad:
        k1 = min(k0 + 64, cpad)
        out.append((k0, k1))
        k0 = k1
    return out


def conv2d_auto(x, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, padding: int = 1,
                act: int = ACT_NONE, slope: float = 0.0, res: Optional[torch.Tensor] = None,
                alpha: float = 1.0, res_mode: int = RES_ADD) -> torch.Tensor:
    """alpha * act(nn.Conv2d(x)) + res on an NHWC bf16 tensor with any channel count.  A 3x3 / stride 1 / pad 1 layer with
    at most 64 input and output channels is one launch of the register-resident MFMA kernel; every other layer with kernel 1
    or 3, stride 1 or 2, padding 0 or 1 is ONE launch of the implicit-GEMM kernel (``conv_gemm``: fp32 accumulation over the
    whole reduction, concatenated sources read in place).  The older composition below remains for A/B timing:

    * more than 64 input channels: one launch per 64-channel slice (read in place), the partial sums accumulated in
      place through the residual input (pre-activation form; partial sums are rounded to bf16 between launches);
    * more than 64 output channels: one launch per 64 output channels, each writing its slice of the result in place;
    * 3x3 / stride 2 / pad 1: the stride-1 layer followed by a 2x2 pick of its even pixels (same arithmetic per
      output pixel, 4x the MFMA work, still far cheaper than the vector-ALU kernel);
    * 1x1 / pad 0: the weight sits in the centre tap of a 3x3 kernel of zeros.

    Everything else (other kernel sizes, stride 2 without padding, ...) runs on the direct kernel.
    x carries pad16(cin) channels; the result carries pad16(cout).  `weight` / `bias` may be Parameters: their
    packed forms are cached per tensor object."""
    cout, cin, kh, kw = weight.shape
    # x may be a tuple of maps: the layer then reads their channel concatenation (torch.concat(..., dim=1) in front of a
    # convolution, core/models/model_final.py:298,375-379,424) without materialising it -- every K-slice is read from
    # the map that holds it
    # (a source may also be ``(map, first_channel, channels)``: a channel window of a wider map, read in place)
    is_window = lambda s: isinstance(s, tuple) and len(s) == 3 and isinstance(s[1], int)
    srcs = list(x) if isinstance(x, (tuple, list)) and not is_window(x) else [x]
    windows = any(is_window(s) for s in srcs)
    if windows:
        gemm_form = use_conv_gemm and weight.shape[2] == weight.shape[3] and weight.shape[2] in (1, 3) and stride in (1, 2) \
            and padding in (0, 1) and padding <= weight.shape[2] // 2 and alpha > 0.0 and len(srcs) <= 3 \
            and all((s[2] if is_window(s) else s.shape[-1]) % 16 == 0 and (s[1] if is_window(s) else 0) % 8 == 0 for s in srcs)
        if gemm_form:
            cpad = sum(s[2] if is_window(s) else s.shape[-1] for s in srcs)
            if cpad - cin >= 16 or cpad < cin:
                raise ValueError("conv2d_auto: the sources' channels must add up to the layer's input channels")
            wp = _cached_pack(weight, ("gemm", cpad), lambda: conv_gemm_pack_weights(weight.detach().float().contiguous(), cpad))
            return conv_gemm(tuple(srcs), wp, None if bias is None else bias.detach(), cout, kh, stride=stride, padding=padding,
                             act=act, slope=slope, alpha=alpha, res=res, res_mode=res_mode)
        srcs = [s[0][..., s[1]:s[1] + s[2]].contiguous() if is_window(s) else s for s in srcs]   # every other path: copies
    if len(srcs) > 1:
        if any(t.shape[:3] != srcs[0].shape[:3] for t in srcs) or sum(t.shape[-1] for t in srcs) - cin >= 16:
            raise ValueError("conv2d_auto: concatenated sources must share (N,H,W) and carry no channel padding except the last")
    cpad = sum(t.shape[-1] for t in srcs)
    as3x3 = (kh, kw, padding) == (3, 3, 1) and stride in (1, 2)
    as1x1 = (kh, kw, padding, stride) == (1, 1, 0, 1)
    gemm_ok = use_conv_gemm and kh == kw and kh in (1, 3) and stride in (1, 2) and padding in (0, 1) and padding <= kh // 2 \
        and alpha > 0.0 and len(srcs) <= 3 and all(t.shape[-1] % 16 == 0 for t in srcs)
    # a layer that is ONE launch of the register-resident kernel stays there (3x3 stride 1, or 1x1 as its centre tap: on
    # the small ESA layers of RepRFN the implicit-GEMM kernel's staging costs more than the eight idle taps)
    one_native = (as3x3 or as1x1) and stride == 1 and len(srcs) == 1 and cpad <= 64 and cout <= 64 and (alpha == 1.0 or as3x3)
    if gemm_ok and not one_native:
        wp = _cached_pack(weight, ("gemm", cpad), lambda: conv_gemm_pack_weights(weight.detach().float().contiguous(), cpad))
        return conv_gemm(tuple(srcs), wp, None if bias is None else bias.detach(), cout, kh, stride=stride, padding=padding,
                         act=act, slope=slope, alpha=alpha, res=res, res_mode=res_mode)
    if res_mode != RES_ADD and not (as3x3 and stride == 1 and len(srcs) == 1 and cpad <= 64 and cout <= 64):
        # the mask form exists on the implicit-GEMM kernel and on the single register-resident launch; elsewhere: a pass
        out = conv2d_auto(x, weight, bias, stride=stride, padding=padding, act=act, slope=slope, alpha=alpha)
        return relu_mask(out, res)
    if not (as3x3 or as1x1) or (alpha != 1.0 and (cpad > 64 or cout > 64 or stride != 1)):
        xin = srcs[0]
        for t in srcs[1:]:
            xin = concat_nhwc(xin, t)
        return conv2d_direct(xin, weight.detach(), None if bias is None else bias.detach(), cin, stride=stride,
                             padding=padding, res=res, act=act, slope=slope, alpha=alpha)
    wd = weight.detach()
    bd = None if bias is None else bias.detach()
    # (source, first channel inside it, weight channel range); slices that hold only channel padding are dropped
    slices, wk = [], 0
    for t in srcs:
        for (a, b) in _k_slices(t.shape[-1]):
            if min(wk + b, cin) > wk + a:
                slices.append((t, a, wk + a, wk + b))
        wk += t.shape[-1]
    n_, h_, w_ = srcs[0].shape[:3]

    def packed(k0, k1, off, n):
        def build():
            w = wd[:, k0:min(k1, cin)]
            if as1x1:
                w3 = torch.zeros((cout, w.shape[1], 3, 3), dtype=torch.float32, device=w.device)
                w3[:, :, 1, 1] = w[:, :, 0, 0]
                w = w3
            return conv3x3_pack_weights(w.float().contiguous(), w.shape[1], k1 - k0, cout=n, cout_off=off)
        return _cached_pack(weight, (k0, k1, off, n, as1x1), build)

    # every launch reads its 64-channel input slice in place and writes (accumulates) its 64-channel output slice in
    # place: no slice copies, no concat
    fuse_res = res is not None and stride == 1 and len(slices) == 1 and cout <= 64
    out = torch.empty((n_, h_, w_, pad16(cout)), dtype=torch.bfloat16, device=srcs[0].device)
    for off in range(0, cout, 64):
        n = min(64, cout - off)
        b = None if bd is None else bd[off:off + n].contiguous()
        for i, (src, a, k0, k1) in enumerate(slices):
            last = i == len(slices) - 1
            wp = packed(k0, k1, off, n)
            sl = None if (a == 0 and k1 - k0 == src.shape[-1]) else (a, k1 - k0)
            if i == 0:
                conv3x3(src, wp, b, n, res=res if (fuse_res and last) else None, act=act if last else ACT_NONE, slope=slope,
                        alpha=alpha, ca_slice=sl, out=out, out_off=off, res_mode=res_mode if (fuse_res and last) else RES_ADD)
            else:
                conv3x3(src, wp, None, n, res=out, res_mode=RES_PRE, act=act if last else ACT_NONE, slope=slope, ca_slice=sl,
                        out=out, out_off=off)
    if stride == 2:
        out = max_pool2d_nhwc(out, 1, 2)  # out[i, j] = stride-1 result at (2i, 2j)
    if res is not None and not fuse_res:
        out = add_bf16(out, res)
    return out


def conv_gemm_pack_weights(weight: torch.Tensor, cin_padded: int) -> torch.Tensor:
    """fp32 (O, I, k, k) master weights, k in {1, 3} -> packed bf16 fragments of the implicit-GEMM convolution."""
    weight = _req(weight, torch.float32, "weight")
    cout, cin, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("conv_gemm_pack_weights: kernel must be 1x1 or 3x3")
    nbytes = lib.srhip_conv_gemm_packed_bytes(cin_padded, cout, kh)
    if nbytes == 0 or cin_padded < cin:
        raise ValueError("conv_gemm_pack_weights: cin_padded must be a multiple of 16 and >= the weight's input channels")
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    check(lib.srhip_conv_gemm_pack_weights(weight.data_ptr(), packed.data_ptr(), cout, cin, kh, cin_padded, _stream()))
    return packed


def conv_gemm_pack_weights_transposed(weight: torch.Tensor, row0: int, rows: int, cin_padded: int) -> torch.Tensor:
    """Packed fragments of the data-gradient layer of a stored fp32 (O, I, k, k) weight: what
    ``conv_gemm_pack_weights(weight.transpose(0, 1).flip(2, 3)[row0:row0 + rows], cin_padded)`` gives, in one kernel."""
    weight = _req(weight, torch.float32, "weight")
    w_o, w_i, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("conv_gemm_pack_weights_transposed: kernel must be 1x1 or 3x3")
    if row0 < 0 or rows < 1 or row0 + rows > w_i:
        raise ValueError("conv_gemm_pack_weights_transposed: rows outside the weight's input channels")
    nbytes = lib.srhip_conv_gemm_packed_bytes(cin_padded, rows, kh)
    if nbytes == 0 or cin_padded < w_o:
        raise ValueError("conv_gemm_pack_weights_transposed: cin_padded must be a multiple of 16 and >= the weight's output channels")
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    check(lib.srhip_conv_gemm_pack_weights_transposed(weight.data_ptr(), packed.data_ptr(), w_o, w_i, kh, row0, rows, cin_padded, _stream()))
    return packed


def conv_gemm(srcs, wpack: torch.Tensor, bias: Optional[torch.Tensor], cout: int, ksize: int, stride: int = 1, padding: int = 0,
              act: int = ACT_NONE, slope: float = 0.0, alpha: float = 1.0, res: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, res_mode: int = RES_ADD) -> torch.Tensor:
    """alpha * act(conv_{k, stride, padding}(concat(srcs)) + bias) + res in ONE launch with fp32 accumulation over the whole
    reduction.  ``srcs``: one NHWC bf16 map or a tuple of up to three with equal (N,H,W), each a multiple of 16 channels wide
    (a source may be ``(tensor, first_channel, channels)`` to read a channel slice in place)."""
    srcs = list(srcs) if isinstance(srcs, (tuple, list)) and not (len(srcs) == 3 and isinstance(srcs[1], int)) else [srcs]
    if not 1 <= len(srcs) <= 3:
        raise ValueError("conv_gemm: one to three sources")
    a = ConvGemmArgs()
    shape = None
    for k, s_ in enumerate(srcs):
        t, off, c = s_ if isinstance(s_, tuple) else (s_, 0, s_.shape[-1])
        t = _req(t, torch.bfloat16, f"src[{k}]")
        if shape is None:
            shape = tuple(t.shape[:3])
        if tuple(t.shape[:3]) != shape or c % 16 or off % 8 or off < 0 or off + c > t.shape[-1]:
            raise ValueError("conv_gemm: sources must share (N,H,W); channel windows are multiples of 16 at offsets that are multiples of 8")
        a.src[k], a.c[k], a.cstride[k] = t.data_ptr() + 2 * off, c, t.shape[-1]
    a.nsrc = len(srcs)
    n, h, w = shape
    ho, wo = (h + 2 * padding - ksize) // stride + 1, (w + 2 * padding - ksize) // stride + 1
    if bias is not None:
        bias = _req(bias, torch.float32, "bias")
    if out is None:
        out = torch.empty((n, ho, wo, pad16(cout)), dtype=torch.bfloat16, device=wpack.device)
    elif out.dtype != torch.bfloat16 or not out.is_contiguous() or tuple(out.shape[:3]) != (n, ho, wo) or out.shape[3] < pad16(cout):
        raise ValueError("conv_gemm: `out` must be a contiguous NHWC bf16 tensor of the output size with at least pad16(cout) channels")
    if res is not None:
        res = _req(res, torch.bfloat16, "res")
        if res.shape != out.shape:
            raise ValueError("conv_gemm: residual shape must equal the output shape")
    a.wpack, a.bias, a.res, a.dst = wpack.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr()
    a.n, a.h, a.w, a.cout, a.cout_stride = n, h, w, cout, out.shape[3]
    a.ksize, a.stride, a.pad, a.act, a.slope, a.alpha = ksize, stride, padding, act, slope, alpha
    a.res_mode = res_mode      # RES_MASK: `res` is the ReLU output in front of a data-gradient launch (zero where it is <= 0)
    check(lib.srhip_conv_gemm_fwd(a, _stream()))
    return out


def conv1x1_chain_supported(channels, has_res: bool) -> bool:
    """Whether ``conv1x1_chain`` has a kernel for this chain of channel counts (input, after each layer)."""
    if len(channels) != 4 or os.environ.get("SRHIP_NO_CHAIN", "0") == "1":
        return False
    import ctypes
    return bool(lib.srhip_conv1x1_chain_supported((ctypes.c_int * 4)(*channels), 3, int(has_res)))


def conv1x1_chain(x: torch.Tensor, layers, alpha: float = 1.0, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """alpha * act2(W2 act1(W1 act0(W0 x + b0) + b1) + b2) + res: three 1x1 convolutions in one pass, the maps between them
    kept on chip (rounded to bf16 as the layer-by-layer form stores them).  ``layers``: three (weight (O,I,1,1) fp32, bias or
    None, act) with channel counts that are multiples of 16; x (N,H,W,I0) bf16, the result (N,H,W,O2)."""
    x = _req(x, torch.bfloat16, "x")
    if len(layers) != 3:
        raise ValueError("conv1x1_chain: three layers")
    a = Conv1x1ChainArgs()
    chans = [x.shape[-1]]
    keep = []
    for l, (wt, b, act) in enumerate(layers):
        cout, cin = wt.shape[0], wt.shape[1]
        if tuple(wt.shape[2:]) != (1, 1) or cin != chans[-1] or cout % 16 or cin % 16:
            raise ValueError("conv1x1_chain: 1x1 kernels whose channel counts chain up, all multiples of 16")
        wp = _cached_pack(wt, ("gemm", cin), lambda wt=wt, cin=cin: conv_gemm_pack_weights(wt.detach().float().contiguous(), cin))
        bb = None if b is None else _req(b.detach(), torch.float32, "bias")
        keep.append((wp, bb))
        a.wpack[l], a.bias[l], a.act[l] = wp.data_ptr(), _ptr(bb), act
        chans.append(cout)
    if not conv1x1_chain_supported(chans, res is not None):
        raise ValueError(f"conv1x1_chain: no kernel for the chain {chans} {'with' if res is not None else 'without'} residual")
    out = torch.empty(x.shape[:-1] + (chans[-1],), dtype=torch.bfloat16, device=x.device)
    if res is not None:
        res = _req(res, torch.bfloat16, "res")
        if res.shape != out.shape:
            raise ValueError("conv1x1_chain: residual shape must equal the output shape")
    a.src, a.res, a.dst = x.data_ptr(), _ptr(res), out.data_ptr()
    a.npix = x.shape[0] * x.shape[1] * x.shape[2]
    for i in range(4):
        a.c[i] = chans[i]
    a.nlayers, a.slope, a.alpha = 3, 0.0, alpha
    check(lib.srhip_conv1x1_chain_fwd(a, _stream()))
    return out


def max_pool2d_nhwc(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    x = _req(x, torch.bfloat16, "x")
    n, h, w, c = x.shape
    out = torch.empty((n, (h - k) // s + 1, (w - k) // s + 1, c), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_max_pool2d_nhwc(x.data_ptr(), out.data_ptr(), n, c, h, w, k, s, _stream()))
    return out


def resize_bilinear_nhwc(x: torch.Tensor, ho: int, wo: int, addend: Optional[torch.Tensor] = None,
                         align_corners: bool = False) -> torch.Tensor:
    """F.interpolate(x, (ho, wo), mode="bilinear", align_corners=align_corners) (+ addend) on NHWC bf16."""
    x = _req(x, torch.bfloat16, "x")
    n, h, w, c = x.shape
    if addend is not None:
        addend = _req(addend, torch.bfloat16, "addend")
        if tuple(addend.shape) != (n, ho, wo, c):
            raise ValueError("resize_bilinear_nhwc: addend must have the output shape")
    out = torch.empty((n, ho, wo, c), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_resize_bilinear_nhwc(x.data_ptr(), _ptr(addend), out.data_ptr(), n, c, h, w, ho, wo, int(align_corners), _stream()))
    return out


def sigmoid_gate(x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    x = _req(x, torch.bfloat16, "x")
    gate = _req(gate, torch.bfloat16, "gate")
    if x.shape != gate.shape:
        raise ValueError("sigmoid_gate: shapes differ")
    out = torch.empty_like(x)
    check(lib.srhip_sigmoid_gate_bf16(x.data_ptr(), gate.data_ptr(), out.data_ptr(), x.numel(), _stream()))
    return out


def esa_tail(x: torch.Tensor, c1: torch.Tensor, c3: torch.Tensor, wf: torch.Tensor, bf: torch.Tensor, w4: torch.Tensor,
             b4: torch.Tensor, cfeat: int) -> torch.Tensor:
    """x * sigmoid(conv4(resize_bilinear(c3 -> x's size) + conv_f(c1))) in one pass (core/arch/RepRFN.py:303-308).
    x (N,h,w,C) bf16 with C >= cfeat, c1 (N,h,w,16), c3 (N,h3,w3,16) bf16; wf (16,16), w4 (cfeat,16) fp32 (1x1 kernels,
    trailing unit dimensions allowed) and their biases."""
    x, c1, c3 = _req(x, torch.bfloat16, "x"), _req(c1, torch.bfloat16, "c1"), _req(c3, torch.bfloat16, "c3")
    wf = _req(wf.reshape(wf.shape[0], -1), torch.float32, "wf")
    w4 = _req(w4.reshape(w4.shape[0], -1), torch.float32, "w4")
    bf, b4 = _req(bf, torch.float32, "bf"), _req(b4, torch.float32, "b4")
    n, h, w, cs = x.shape
    if c1.shape != (n, h, w, 16) or c3.shape[0] != n or c3.shape[3] != 16 or tuple(wf.shape) != (16, 16) or tuple(w4.shape) != (cfeat, 16) \
            or bf.numel() != 16 or b4.numel() != cfeat:
        raise ValueError("esa_tail: c1 (N,h,w,16), c3 (N,h3,w3,16), wf (16,16), w4 (cfeat,16)")
    out = torch.empty_like(x)   # channels beyond cfeat are copied from x
    check(lib.srhip_esa_tail_bf16(x.data_ptr(), c1.data_ptr(), c3.data_ptr(), wf.data_ptr(), bf.data_ptr(), w4.data_ptr(),
                                  b4.data_ptr(), out.data_ptr(), n, h, w, c3.shape[1], c3.shape[2], cfeat, cs, _stream()))
    return out


def l1_ssim_loss(sr: torch.Tensor, target: torch.Tensor, lambda_l1: float, lambda_ssim: float,
                 clip_min: float = float("-inf"), clip_max: float = float("inf")):
    """Fused clip + L1 + SSIM loss and its gradient: returns (out3, dsr) with out3 = [l1, ssim_loss, total]
    (device tensor) and dsr = d total / d sr."""
    sr = _req(sr, torch.float32, "sr")
    target = _req(target, torch.float32, "target")
    if sr.shape != target.shape or sr.dim() != 4:
        raise ValueError("l1_ssim_loss: sr and target must be equal-shape (N,C,H,W) tensors")
    n, c, h, w = sr.shape
    ws = torch.empty((lib.srhip_l1_ssim_workspace_bytes(n * c, h, w),), dtype=torch.uint8, device=sr.device)
    dsr = torch.empty_like(sr)
    out3 = torch.empty((3,), dtype=torch.float32, device=sr.device)
    check(lib.srhip_l1_ssim_loss(sr.data_ptr(), target.data_ptr(), dsr.data_ptr(), out3.data_ptr(), ws.data_ptr(), n * c, h, w,
                                 lambda_l1, lambda_ssim, clip_min, clip_max, _stream()))
    return out3, dsr


def weighted_l1_loss(sr: torch.Tensor, target: torch.Tensor, weight: torch.Tensor):
    """mean((1 + weight, nearest-upsampled and broadcast over the channels) * |sr - target|) and its gradient:
    returns (loss (1,) device tensor, dsr).  sr / target (N,C,H,W) fp32, weight (N,1,h,w) fp32 with h | H, w | W."""
    sr = _req(sr, torch.float32, "sr")
    target = _req(target, torch.float32, "target")
    weight = _req(weight, torch.float32, "weight")
    if sr.shape != target.shape or sr.dim() != 4:
        raise ValueError("weighted_l1_loss: sr and target must be equal-shape (N,C,H,W) tensors")
    n, c, h, w = sr.shape
    if weight.dim() != 4 or weight.shape[0] != n or weight.shape[1] != 1:
        raise ValueError("weighted_l1_loss: weight must be (N,1,h,w)")
    ws = torch.empty((lib.srhip_weighted_l1_workspace_bytes(),), dtype=torch.uint8, device=sr.device)
    dsr = torch.empty_like(sr)
    out = torch.empty((1,), dtype=torch.float32, device=sr.device)
    check(lib.srhip_weighted_l1_loss(sr.data_ptr(), target.data_ptr(), weight.data_ptr(), dsr.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                     n, c, h, w, weight.shape[2], weight.shape[3], _stream()))
    return out, dsr


def conv2d_direct(src: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], cin: int, stride=1, padding=0,
                  res: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0, alpha: float = 1.0,
                  cout_stride: Optional[int] = None) -> torch.Tensor:
    """Generic NHWC bf16 convolution (fp32 OIHW weights) for small / odd-shaped layers."""
    src = _req(src, torch.bfloat16, "src")
    weight = _req(weight, torch.float32, "weight")
    n, h, w, cs = src.shape
    cout, wcin, kh, kw = weight.shape
    if wcin != cin:
        raise ValueError("conv2d_direct: weight C_in does not match cin")
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    ho, wo = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    cout_stride = pad16(cout) if cout_stride is None else cout_stride
    out = torch.empty((n, ho, wo, cout_stride), dtype=torch.bfloat16, device=src.device)
    if bias is not None:
        bias = _req(bias, torch.float32, "bias")
    if res is not None:
        res = _req(res, torch.bfloat16, "res")
    check(lib.srhip_conv2d_direct(src.data_ptr(), weight.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr(), n, h, w, cin,
                                  cs, cout, cout_stride, kh, kw, sh, sw, ph, pw, act, slope, alpha, _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# memory-bound operators
# ------------------------------------------------------------------------------------------------
def upsample_grid_bilinear_nhwc(x: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    x = _req(x, torch.bfloat16, "x")
    n, sh, sw, c = x.shape
    out = torch.empty((n, th, tw, c), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_upsample_grid_bilinear_nhwc(x.data_ptr(), out.data_ptr(), n, c, sh, sw, th, tw, _stream()))
    return out


def upsample_grid_bilinear(x: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    """F.grid_sample(x, make_coord((th, tw)), bilinear, zeros, align_corners=False) on fp32 NCHW."""
    x = _req(x, torch.float32, "x")
    n, c, sh, sw = x.shape
    out = torch.empty((n, c, th, tw), dtype=torch.float32, device=x.device)
    check(lib.srhip_upsample_grid_bilinear_f32(x.data_ptr(), out.data_ptr(), n, c, sh, sw, th, tw, _stream()))
    return out


def upsample_grid_bilinear_bwd(dy: torch.Tensor, sh: int, sw: int) -> torch.Tensor:
    dy = _req(dy, torch.float32, "dy")
    n, c, th, tw = dy.shape
    dx = torch.empty((n, c, sh, sw), dtype=torch.float32, device=dy.device)
    check(lib.srhip_upsample_grid_bilinear_bwd_f32(dy.data_ptr(), dx.data_ptr(), n, c, sh, sw, th, tw, _stream()))
    return dx


def backward_warp(frame: torch.Tensor, motion: torch.Tensor, mode: str = "nearest", align_corners: bool = True) -> torch.Tensor:
    """fp32 NCHW motion-vector warp (semantics of the reference's backward_warping)."""
    frame = _req(frame, torch.float32, "frame")
    motion = _req(motion, torch.float32, "motion")
    n, c, h, w = frame.shape
    if tuple(motion.shape) != (n, 2, h, w):
        raise ValueError("backward_warp: motion must be (N,2,H,W)")
    out = torch.empty_like(frame)
    check(lib.srhip_backward_warp_f32(frame.data_ptr(), motion.data_ptr(), out.data_ptr(), n, c, h, w, _MODES[mode],
                                      int(align_corners), _stream()))
    return out


def warp_concat_nhwc(buffers: torch.Tensor, prev: Optional[torch.Tensor], motion: Optional[torch.Tensor],
                     mode: str = "bilinear", align_corners: bool = True) -> torch.Tensor:
    """(N,cb,h,w) fp32 G-buffers + the previous output (N,cf,h,w) warped by motion (N,2,h,w) -> (N,h,w,16) bf16 NHWC,
    channels [buffers | warped previous frame | zeros], in one pass (no fp32 warp result, no fp32 concatenation).
    prev None: the warped channels are zero."""
    buffers = _req(buffers, torch.float32, "buffers")
    n, cb, h, w = buffers.shape
    cf = 0
    if prev is not None:
        prev = _req(prev, torch.float32, "prev")
        if motion is None:
            raise ValueError("warp_concat_nhwc: a previous frame needs motion vectors")
        motion = _req(motion, torch.float32, "motion")
        cf = prev.shape[1]
        if tuple(prev.shape) != (n, cf, h, w) or tuple(motion.shape) != (n, 2, h, w):
            raise ValueError("warp_concat_nhwc: prev must be (N,cf,H,W) and motion (N,2,H,W) at the buffers' size")
    if cb + cf > 16:
        raise ValueError(f"warp_concat_nhwc: {cb} + {cf} channels do not fit the 16-channel pixel")
    out = torch.empty((n, h, w, 16), dtype=torch.bfloat16, device=buffers.device)
    check(lib.srhip_warp_concat_nchw_f32_to_nhwc_bf16(buffers.data_ptr(), cb, prev.data_ptr() if prev is not None else None, cf,
                                                      motion.data_ptr() if prev is not None else None, out.data_ptr(), n, h, w,
                                                      _MODES[mode], int(align_corners), _stream()))
    return out


def backward_warp_nhwc(frame: torch.Tensor, motion: torch.Tensor, mode: str = "bilinear", align_corners: bool = True) -> torch.Tensor:
    frame = _req(frame, torch.bfloat16, "frame")
    motion = _req(motion, torch.float32, "motion")
    n, h, w, c = frame.shape
    if tuple(motion.shape) != (n, 2, h, w):
        raise ValueError("backward_warp_nhwc: motion must be (N,2,H,W)")
    out = torch.empty_like(frame)
    check(lib.srhip_backward_warp_nhwc(frame.data_ptr(), motion.data_ptr(), out.data_ptr(), n, c, h, w, _MODES[mode],
                                       int(align_corners), _stream()))
    return out


def backward_warp_bwd(frame: torch.Tensor, motion: torch.Tensor, dy: torch.Tensor, align_corners: bool = True):
    frame = _req(frame, torch.float32, "frame")
    motion = _req(motion, torch.float32, "motion")
    dy = _req(dy, torch.float32, "dy")
    n, c, h, w = frame.shape
    dframe = torch.empty_like(frame)
    dmotion = torch.empty_like(motion)
    check(lib.srhip_backward_warp_bwd_f32(frame.data_ptr(), motion.data_ptr(), dy.data_ptr(), dframe.data_ptr(),
                                          dmotion.data_ptr(), n, c, h, w, int(align_corners), _stream()))
    return dframe, dmotion


def _elem(t: torch.Tensor) -> int:
    if t.element_size() not in (2, 4):
        raise TypeError("expected a 2- or 4-byte element type")
    return t.element_size()


def pixel_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    if not x.is_cuda:
        raise ValueError("pixel_shuffle: expected a GPU tensor")
    x = x.contiguous()
    n, crr, h, w = x.shape
    if crr % (r * r):
        raise ValueError("pixel_shuffle: channels not divisible by r*r")
    c = crr // (r * r)
    out = torch.empty((n, c, h * r, w * r), dtype=x.dtype, device=x.device)
    check(lib.srhip_pixel_shuffle(x.data_ptr(), out.data_ptr(), n, c, h, w, r, _elem(x), _stream()))
    return out


def pixel_unshuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    if not x.is_cuda:
        raise ValueError("pixel_unshuffle: expected a GPU tensor")
    x = x.contiguous()
    n, c, hr, wr = x.shape
    if hr % r or wr % r:
        raise ValueError("pixel_unshuffle: spatial size not divisible by r")
    h, w = hr // r, wr // r
    out = torch.empty((n, c * r * r, h, w), dtype=x.dtype, device=x.device)
    check(lib.srhip_pixel_unshuffle(x.data_ptr(), out.data_ptr(), n, c, h, w, r, _elem(x), _stream()))
    return out


def pixel_unshuffle_to_nhwc_bf16(x: torch.Tensor, r: int, want_max: bool = False):
    """nchw_f32_to_nhwc_bf16(pixel_unshuffle(x, r)) -- one kernel for r = 4 and up to 23 channels, the pair otherwise.
    want_max: also return x.max() as a 0-d device tensor (NaN if x holds one), found by the same pass over x."""
    x = _req(x, torch.float32, "x")
    n, c, hr, wr = x.shape
    if hr % r or wr % r:
        raise ValueError("pixel_unshuffle: spatial size not divisible by r")
    if r != 4 or c > 23:
        out = nchw_f32_to_nhwc_bf16(pixel_unshuffle(x, r))
        return (out, x.max()) if want_max else out
    out = torch.empty((n, hr // 4, wr // 4, 16 * c), dtype=torch.bfloat16, device=x.device)
    if want_max:
        part = torch.empty(4096, dtype=torch.float32, device=x.device)
        check(lib.srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max(x.data_ptr(), out.data_ptr(), n, c, hr // 4, wr // 4,
                                                                   part.data_ptr(), part.numel(), _stream()))
        return out, part.max()
    check(lib.srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), n, c, hr // 4, wr // 4, _stream()))
    return out


def gradient_map_unshuffle_nhwc(x: torch.Tensor, planes, eps: float = 1e-6) -> torch.Tensor:
    """nchw_f32_to_nhwc_bf16(pixel_unshuffle(gradient_map(x[:, planes]), 4)) in one pass (GradientMap + PixelUnshuffle of
    core/models/model_final.py:406-409): x (N,C,4h,4w) fp32 -> (N,h,w,16*len(planes)) bf16."""
    x = _req(x, torch.float32, "x")
    n, cs, hr, wr = x.shape
    planes = [int(k) for k in planes]
    if hr % 4 or wr % 4:
        raise ValueError("gradient_map_unshuffle_nhwc: spatial size not divisible by 4")
    if not 1 <= len(planes) <= 23 or any(k < 0 or k >= cs for k in planes):
        raise ValueError("gradient_map_unshuffle_nhwc: 1 to 23 plane indices inside the tensor")
    out = torch.empty((n, hr // 4, wr // 4, 16 * len(planes)), dtype=torch.bfloat16, device=x.device)
    idx = (ctypes.c_int * len(planes))(*planes)
    check(lib.srhip_gradient_map_unshuffle4_nhwc_bf16(x.data_ptr(), cs, idx, len(planes), out.data_ptr(), n, hr // 4, wr // 4, eps, _stream()))
    return out


def pixel_shuffle4_nhwc_to_nhwc16(x: torch.Tensor, c: int) -> torch.Tensor:
    """PixelShuffle(4) of an NHWC bf16 map with at least 16c channels, written as the 16-channel NHWC bf16 input of the next
    convolution (channels >= c zero): (N,h,w,cs) -> (N,4h,4w,16).  c <= 4."""
    x = _req(x, torch.bfloat16, "x")
    n, h, w, cs = x.shape
    if not 1 <= c <= 4 or cs < 16 * c:
        raise ValueError("pixel_shuffle4_nhwc_to_nhwc16: 1 to 4 output channels out of at least 16 c input channels")
    out = torch.empty((n, 4 * h, 4 * w, 16), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_pixel_shuffle4_nhwc_bf16_to_nhwc16(x.data_ptr(), out.data_ptr(), n, c, h, w, cs, _stream()))
    return out


def zero_insert2_nhwc(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """(N,ho,wo,C) bf16 -> (N,h,w,C) with x at the even positions and zeros elsewhere (ho = ceil(h/2), wo = ceil(w/2))."""
    x = _req(x, torch.bfloat16, "x")
    n, ho, wo, c = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=x.device)
    check(lib.srhip_zero_insert2_nhwc_bf16(x.data_ptr(), out.data_ptr(), n, ho, wo, h, w, c, _stream()))
    return out


def pixel_shuffle4_nhwc16_bwd(d: torch.Tensor, c: int, cs: int) -> torch.Tensor:
    """Backward of pixel_shuffle4_nhwc_to_nhwc16: d (N,4h,4w,16) bf16 -> (N,h,w,cs) bf16 (zero outside the first 16c channels)."""
    d = _req(d, torch.bfloat16, "d")
    n, hh, ww, c16 = d.shape
    if c16 != 16 or hh % 4 or ww % 4 or not 1 <= c <= 4 or cs < 16 * c or cs % 8:
        raise ValueError("pixel_shuffle4_nhwc16_bwd: d must be (N,4h,4w,16), 1 <= c <= 4, cs >= 16 c a multiple of 8")
    out = torch.empty((n, hh // 4, ww // 4, cs), dtype=torch.bfloat16, device=d.device)
    check(lib.srhip_pixel_shuffle4_nhwc16_bwd_bf16(d.data_ptr(), out.data_ptr(), n, c, hh // 4, ww // 4, cs, _stream()))
    return out


def pixel_shuffle4_conv3x3_c3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE,
                              slope: float = 0.0) -> torch.Tensor:
    """act(conv2d(pixel_shuffle(x, 4), weight, bias, padding=1)) for a 3 -> 3 channel 3x3 layer, in one pass:
    x (N,h,w,>=48) bf16 NHWC -> (N,3,4h,4w) fp32 NCHW; fp32 weights and arithmetic."""
    x = _req(x, torch.bfloat16, "x")
    weight = _req(weight, torch.float32, "weight")
    if tuple(weight.shape) != (3, 3, 3, 3) or x.shape[-1] < 48:
        raise ValueError("pixel_shuffle4_conv3x3_c3: a (3,3,3,3) weight and at least 48 input channels")
    if bias is not None:
        bias = _req(bias, torch.float32, "bias")
        if bias.numel() != 3:
            raise ValueError("pixel_shuffle4_conv3x3_c3: bias must have 3 elements")
    n, h, w, cs = x.shape
    out = torch.empty((n, 3, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
    check(lib.srhip_pixel_shuffle4_conv3x3_c3_nchw_f32(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), n, h, w, cs,
                                                       act, slope, _stream()))
    return out


def pixel_shuffle_nhwc_to_nchw_f32(x: torch.Tensor, c: int, r: int) -> torch.Tensor:
    x = _req(x, torch.bfloat16, "x")
    n, h, w, cs = x.shape
    out = torch.empty((n, c, h * r, w * r), dtype=torch.float32, device=x.device)
    check(lib.srhip_pixel_shuffle_nhwc_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, r, cs, _stream()))
    return out


def nearest_down(x: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """F.interpolate(x, scale_factor=scale_factor, mode="nearest") for scale_factor <= 1 (NCHW)."""
    if not x.is_cuda:
        raise ValueError("nearest_down: expected a GPU tensor")
    x = x.contiguous()
    n, c, h, w = x.shape
    ho, wo = int(h * scale_factor), int(w * scale_factor)
    out = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device)
    check(lib.srhip_nearest_down(x.data_ptr(), out.data_ptr(), n * c, h, w, ho, wo, float(scale_factor), _elem(x), _stream()))
    return out


def rel_coord(th: int, tw: int, sh: int, sw: int, device) -> torch.Tensor:
    out = torch.empty((1, 2, th, tw), dtype=torch.float32, device=device)
    check(lib.srhip_rel_coord(out.data_ptr(), th, tw, sh, sw, _stream()))
    return out


def make_coord(h: int, w: int, device) -> torch.Tensor:
    out = torch.empty((1, 2, h, w), dtype=torch.float32, device=device)
    check(lib.srhip_make_coord(out.data_ptr(), h, w, _stream()))
    return out


def concat_nhwc(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a = _req(a, torch.bfloat16, "a")
    b = _req(b, torch.bfloat16, "b")
    if a.shape[:-1] != b.shape[:-1]:
        raise ValueError("concat_nhwc: leading dimensions differ")
    out = torch.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), dtype=torch.bfloat16, device=a.device)
    pixels = a.numel() // a.shape[-1]
    check(lib.srhip_concat_nhwc(a.data_ptr(), b.data_ptr(), out.data_ptr(), pixels, a.shape[-1], b.shape[-1], _stream()))
    return out


def layernorm_cf(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    x = _req(x, torch.float32, "x")
    weight = _req(weight, torch.float32, "weight")
    bias = _req(bias, torch.float32, "bias")
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    check(lib.srhip_layernorm_cf_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), n, c, h * w, eps, _stream()))
    return y


def layernorm_cf_bwd(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, eps: float = 1e-6):
    """-> (dx, dweight, dbias) of layernorm_cf."""
    x = _req(x, torch.float32, "x")
    weight = _req(weight, torch.float32, "weight")
    dy = _req(dy, torch.float32, "dy")
    if dy.shape != x.shape:
        raise ValueError("layernorm_cf_bwd: dy must have the shape of x")
    n, c, h, w = x.shape
    dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
    ws = torch.empty((lib.srhip_layernorm_cf_bwd_workspace_bytes(n, c, h * w),), dtype=torch.uint8, device=x.device)
    check(lib.srhip_layernorm_cf_bwd_f32(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                         db.data_ptr(), ws.data_ptr(), n, c, h * w, eps, _stream()))
    return dx, dw, db


class _LayerNormCF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return layernorm_cf(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = layernorm_cf_bwd(x, weight, dy.contiguous(), ctx.eps)
        return dx, dw, db, None


def layernorm_cf_autograd(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Differentiable channels-first LayerNorm (core/arch/CAMixerSR.py:29-37 under loss.backward())."""
    return _LayerNormCF.apply(x, weight, bias, eps)


# ------------------------------------------------------------------------------------------------
# temporal variant: disocclusion logic, depth compression
# ------------------------------------------------------------------------------------------------
def disocclusion(cur_frame: torch.Tensor, cur_buffer: torch.Tensor, hist_frames: torch.Tensor, hist_buffer: torch.Tensor,
                 want_mask: bool = False, aux_planes: int = 14):
    """core/models/model_final.py:472-486,501-504 in one kernel -> (marked_frames (B*F,3,h,w), aux (B*F,14,h,w),
    enhance_weight (B,1,h,w)[, mask (B,F,1,h,w) bool]).  aux_planes > 14 (only for B*F == 1, where the 14 planes are a
    contiguous prefix) returns aux as (1,aux_planes,h,w) with the planes behind the fourteenth left for the caller to
    fill: the history refiner's input is these planes followed by |motion| (model_final.py:300)."""
    cur_frame = _req(cur_frame, torch.float32, "cur_frame")
    cur_buffer = _req(cur_buffer, torch.float32, "cur_buffer")
    hist_frames = _req(hist_frames, torch.float32, "hist_frames")
    hist_buffer = _req(hist_buffer, torch.float32, "hist_buffer")
    b, f, cf, h, w = hist_frames.shape
    if cf != 3 or tuple(cur_frame.shape) != (b, 3, h, w) or tuple(cur_buffer.shape) != (b, 10, h, w) \
            or tuple(hist_buffer.shape) != (b, f, 10, h, w):
        raise ValueError("disocclusion: expected cur_frame (B,3,h,w), cur_buffer (B,10,h,w), hist_frames (B,F,3,h,w), "
                         "hist_buffer (B,F,10,h,w)")
    dev = cur_frame.device
    marked = torch.empty((b * f, 3, h, w), dtype=torch.float32, device=dev)
    if aux_planes != 14 and (aux_planes < 14 or b * f != 1):
        raise ValueError("disocclusion: aux_planes > 14 needs B*F == 1")
    aux = torch.empty((b * f, aux_planes, h, w), dtype=torch.float32, device=dev)
    ew = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
    mask = torch.empty((b, f, 1, h, w), dtype=torch.uint8, device=dev) if want_mask else None
    check(lib.srhip_disocclusion_f32(cur_frame.data_ptr(), cur_buffer.data_ptr(), hist_frames.data_ptr(), hist_buffer.data_ptr(),
                                     _ptr(mask), marked.data_ptr(), aux.data_ptr(), ew.data_ptr(), b, f, h, w, _stream()))
    return (marked, aux, ew, mask.bool()) if want_mask else (marked, aux, ew)


def clamp_depth(depth: torch.Tensor, thres1: float = 3.0, thres2: float = 7.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """core/models/model_final.py:40-46 (identity below thres1, so no host-side max() test)."""
    depth = _req(depth, torch.float32, "depth")
    if out is None:
        out = torch.empty_like(depth)
    elif out.dtype != torch.float32 or out.shape != depth.shape or not out.is_contiguous() or out.device != depth.device:
        raise ValueError("clamp_depth: out must be a contiguous fp32 tensor of depth's shape on depth's device (it may be depth itself)")
    check(lib.srhip_clamp_depth_f32(depth.data_ptr(), out.data_ptr(), depth.numel(), thres1, thres2, _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# backward pieces of RepRFN (LeakyReLU from its output, ESA gate / pool / resize)
# ------------------------------------------------------------------------------------------------
def lrelu_mask(dy: torch.Tensor, y: torch.Tensor, slope: float) -> torch.Tensor:
    """dy * (y > 0 ? 1 : slope) (bf16): backward of LeakyReLU(slope) / ReLU (slope 0) from the activation's output."""
    dy = _req(dy, torch.bfloat16, "dy")
    y = _req(y, torch.bfloat16, "y")
    if dy.shape != y.shape:
        raise ValueError("lrelu_mask: shapes differ")
    out = torch.empty_like(dy)
    check(lib.srhip_lrelu_mask_bf16(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), slope, _stream()))
    return out


def sigmoid_gate_bwd(x: torch.Tensor, gate: torch.Tensor, dy: torch.Tensor):
    x, gate, dy = _req(x, torch.bfloat16, "x"), _req(gate, torch.bfloat16, "gate"), _req(dy, torch.bfloat16, "dy")
    if x.shape != gate.shape or x.shape != dy.shape:
        raise ValueError("sigmoid_gate_bwd: shapes differ")
    dx, dg = torch.empty_like(x), torch.empty_like(x)
    check(lib.srhip_sigmoid_gate_bwd_bf16(x.data_ptr(), gate.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), x.numel(), _stream()))
    return dx, dg


def max_pool2d_bwd_nhwc(x: torch.Tensor, dy: torch.Tensor, k: int, s: int) -> torch.Tensor:
    x, dy = _req(x, torch.bfloat16, "x"), _req(dy, torch.bfloat16, "dy")
    n, h, w, c = x.shape
    if tuple(dy.shape) != (n, (h - k) // s + 1, (w - k) // s + 1, c):
        raise ValueError("max_pool2d_bwd_nhwc: dy does not have the pooled shape")
    dx = torch.empty_like(x)
    check(lib.srhip_max_pool2d_bwd_nhwc(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c, h, w, k, s, _stream()))
    return dx


def dilate_velocity(motion: torch.Tensor, depth: torch.Tensor, ksize: int = 3) -> torch.Tensor:
    """core/tools/math.py:80-100: motion (B,H,W,2) fp32, depth (B,1,H,W) fp32 -> (B,H,W,2), the motion vector of the
    nearest-depth pixel of every ksize x ksize neighbourhood."""
    motion, depth = _req(motion, torch.float32, "motion"), _req(depth, torch.float32, "depth")
    if motion.dim() != 4 or motion.shape[-1] != 2:
        raise ValueError("dilate_velocity: motion must be (B,H,W,2)")
    b, h, w, _ = motion.shape
    if tuple(depth.shape) != (b, 1, h, w):
        raise ValueError("dilate_velocity: depth must be (B,1,H,W) with motion's B, H, W")
    out = torch.empty_like(motion)
    check(lib.srhip_dilate_velocity_f32(motion.data_ptr(), depth.data_ptr(), out.data_ptr(), b, h, w, ksize, _stream()))
    return out


def attention_fuse(cur: torch.Tensor, fmap: torch.Tensor, grad: torch.Tensor, f_self: torch.Tensor, g_cross: torch.Tensor,
                   g_self: torch.Tensor, f_cross: torch.Tensor) -> torch.Tensor:
    """cur + sigmoid(f_self * g_cross) * fmap + sigmoid(g_self * f_cross) * grad (core/models/model_final.py:422).
    cur / fmap / grad: (N,H,W,C) bf16 contiguous; the gates: (N,H,W,1) bf16 views of maps with one common channel stride."""
    cur, fmap, grad = _req(cur, torch.bfloat16, "cur"), _req(fmap, torch.bfloat16, "fmap"), _req(grad, torch.bfloat16, "grad")
    if fmap.shape != cur.shape or grad.shape != cur.shape:
        raise ValueError("attention_fuse: cur, fmap and grad must have one shape")
    n, h, w, c = cur.shape
    gates = (f_self, g_cross, g_self, f_cross)
    gs = gates[0].stride(2)
    for t in gates:
        if t.dtype != torch.bfloat16 or not t.is_cuda or tuple(t.shape) != (n, h, w, 1) or t.stride() != (h * w * gs, w * gs, gs, 1):
            raise ValueError("attention_fuse: every gate must be a (N,H,W,1) bf16 view with pixel stride %d" % gs)
    out = torch.empty_like(cur)
    check(lib.srhip_attention_fuse_bf16(cur.data_ptr(), fmap.data_ptr(), grad.data_ptr(), f_self.data_ptr(), g_cross.data_ptr(),
                                        g_self.data_ptr(), f_cross.data_ptr(), gs, out.data_ptr(), n * h * w, c, _stream()))
    return out


def attention_fuse_bwd(d: torch.Tensor, fmap: torch.Tensor, grad: torch.Tensor, f_self: torch.Tensor, g_cross: torch.Tensor,
                       g_self: torch.Tensor, f_cross: torch.Tensor, gate_channels: int = 16):
    """Backward of attention_fuse from d = the gradient of its result (also the gradient of ``cur``): returns (dfmap, dgrad,
    d_f_self, d_g_cross, d_g_self, d_f_cross); the gate gradients are (N,H,W,gate_channels) maps with the value in channel
    0 and zeros elsewhere (the gradient of the padded map whose channel 0 the gate is)."""
    d, fmap, grad = _req(d, torch.bfloat16, "d"), _req(fmap, torch.bfloat16, "fmap"), _req(grad, torch.bfloat16, "grad")
    if fmap.shape != d.shape or grad.shape != d.shape:
        raise ValueError("attention_fuse_bwd: d, fmap and grad must have one shape")
    n, h, w, c = d.shape
    gates = (f_self, g_cross, g_self, f_cross)
    gs = gates[0].stride(2)
    for t in gates:
        if t.dtype != torch.bfloat16 or not t.is_cuda or tuple(t.shape) != (n, h, w, 1) or t.stride() != (h * w * gs, w * gs, gs, 1):
            raise ValueError("attention_fuse_bwd: every gate must be a (N,H,W,1) bf16 view with pixel stride %d" % gs)
    if gate_channels < 8 or gate_channels % 8:
        raise ValueError("attention_fuse_bwd: gate_channels must be a multiple of 8")
    dfmap, dgrad = torch.empty_like(d), torch.empty_like(d)
    dg = [torch.empty((n, h, w, gate_channels), dtype=torch.bfloat16, device=d.device) for _ in range(4)]
    check(lib.srhip_attention_fuse_bwd_bf16(d.data_ptr(), fmap.data_ptr(), grad.data_ptr(), f_self.data_ptr(), g_cross.data_ptr(),
                                            g_self.data_ptr(), f_cross.data_ptr(), gs, dfmap.data_ptr(), dgrad.data_ptr(),
                                            dg[0].data_ptr(), dg[1].data_ptr(), dg[2].data_ptr(), dg[3].data_ptr(), gate_channels,
                                            n * h * w, c, _stream()))
    return (dfmap, dgrad) + tuple(dg)


def max_pool2d_arg_nhwc(x: torch.Tensor, k: int, s: int):
    """max_pool2d_nhwc that also returns the window position of every maximum ((N,ho,wo,C) uint8), for the backward."""
    x = _req(x, torch.bfloat16, "x")
    n, h, w, c = x.shape
    if not 1 <= k <= 15:
        raise ValueError("max_pool2d_arg_nhwc: window size must be 1..15")
    shape = (n, (h - k) // s + 1, (w - k) // s + 1, c)
    out = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
    arg = torch.empty(shape, dtype=torch.uint8, device=x.device)
    check(lib.srhip_max_pool2d_arg_nhwc(x.data_ptr(), out.data_ptr(), arg.data_ptr(), n, c, h, w, k, s, _stream()))
    return out, arg


def max_pool2d_bwd_arg_nhwc(arg: torch.Tensor, dy: torch.Tensor, h: int, w: int, k: int, s: int) -> torch.Tensor:
    """Gradient of the (N,h,w,C) pooled input from the recorded positions and dy (both (N,ho,wo,C))."""
    arg, dy = _req(arg, torch.uint8, "arg"), _req(dy, torch.bfloat16, "dy")
    n, ho, wo, c = dy.shape
    if tuple(arg.shape) != tuple(dy.shape) or (ho, wo) != ((h - k) // s + 1, (w - k) // s + 1):
        raise ValueError("max_pool2d_bwd_arg_nhwc: arg / dy do not have the pooled shape")
    dx = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dy.device)
    check(lib.srhip_max_pool2d_bwd_arg_nhwc(arg.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c, h, w, k, s, _stream()))
    return dx


def resize_bilinear_bwd_nhwc(dy: torch.Tensor, h: int, w: int, align_corners: bool = False) -> torch.Tensor:
    """Adjoint of resize_bilinear_nhwc: dy (N,ho,wo,C) -> gradient of the (N,h,w,C) source."""
    dy = _req(dy, torch.bfloat16, "dy")
    n, ho, wo, c = dy.shape
    dx = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dy.device)
    check(lib.srhip_resize_bilinear_bwd_nhwc(dy.data_ptr(), dx.data_ptr(), n, c, h, w, ho, wo, int(align_corners), _stream()))
    return dx
