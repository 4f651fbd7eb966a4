"""GPU parity tests: every HIP operator is called through the C ABI (via the ctypes wrappers) and
compared with the CPU oracle and with the goldens the reference produced.

Tolerances (from the task statement): bit-exact for pixel shuffle / index math; convolution and
warp max-abs-error <= 2e-2 with bf16 outputs and <= 1e-3 with fp32 accumulate/outputs."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

BF16_TOL = 2e-2
F32_TOL = 1e-3


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16r(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def run_conv3x3(F_, x, w, b, *, split=None, res=None, act=0, slope=0.0, alpha=1.0, f32_out=False):
    """x (N,C,H,W) fp32 (bf16-representable), w OIHW; returns NCHW fp32 numpy."""
    cout, cin = w.shape[:2]
    wd = dev(w)
    if split is None:
        xa = F_.nchw_f32_to_nhwc_bf16(dev(x))
        wp = F_.conv3x3_pack_weights(wd, cin, xa.shape[-1])
        xb = None
    else:
        xa = F_.nchw_f32_to_nhwc_bf16(dev(x[:, :split]))
        xb = F_.nchw_f32_to_nhwc_bf16(dev(x[:, split:]))
        wp = F_.conv3x3_pack_weights(wd, split, xa.shape[-1], cin - split, xb.shape[-1])
    r = None
    if res is not None:
        r = F_.nchw_f32_to_nhwc_bf16(dev(res))
    y = F_.conv3x3(xa, wp, dev(b) if b is not None else None, cout, src_b=xb, res=r, act=act, slope=slope, alpha=alpha,
                   out_nchw_f32=f32_out)
    torch.cuda.synchronize()
    if f32_out:
        return y.cpu().numpy()
    return F_.nhwc_bf16_to_nchw_f32(y, cout).cpu().numpy()


# ------------------------------------------------------------------------------------------------
# layout kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [3, 10, 16, 32, 48, 64])
def test_layout_roundtrip_bit_exact(F_, c):
    rng = np.random.default_rng(c)
    x = bf16r(rng.standard_normal((2, c, 7, 13)))
    y = F_.nchw_f32_to_nhwc_bf16(dev(x))
    assert y.shape == (2, 7, 13, (c + 15) // 16 * 16)
    ref = np.zeros(y.shape, np.float32)
    ref[..., :c] = np.transpose(x, (0, 2, 3, 1))
    assert np.array_equal(y.float().cpu().numpy(), ref)
    back = F_.nhwc_bf16_to_nchw_f32(y, c).cpu().numpy()
    assert np.array_equal(back, x)


@pytest.mark.parametrize("c,shape", [(96, (2, 7, 13)), (112, (1, 16, 16)), (160, (3, 5, 43)), (384, (1, 9, 8)), (100, (2, 1, 1))])
def test_layout_wide_channel_counts_bit_exact(F_, c, shape):
    """Channel counts of the temporal network's unshuffled buffers (core/models/model_final.py:95-105) go through the
    LDS-tiled kernel (64-pixel tiles: partial tiles, tiles that straddle two images) or, above 368 channels, the plain one."""
    n, h, w = shape
    rng = np.random.default_rng(c + h)
    x = bf16r(rng.standard_normal((n, c, h, w)))
    y = F_.nchw_f32_to_nhwc_bf16(dev(x))
    cp = (c + 15) // 16 * 16
    assert y.shape == (n, h, w, cp)
    ref = np.zeros(y.shape, np.float32)
    ref[..., :c] = np.transpose(x, (0, 2, 3, 1))
    assert np.array_equal(y.float().cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------------
# MFMA convolution vs goldens
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,split", [("c64_64", None), ("c10_64", None), ("c96_64", 32), ("c64_3", None),
                                        ("c3_32", None), ("c32_32", None), ("c48_48", None)])
def test_conv3x3_golden(F_, ops_golden, name, split):
    z = ops_golden
    x, w, b, yref = z[f"conv.{name}.x"], z[f"conv.{name}.w"], z[f"conv.{name}.b"], z[f"conv.{name}.y"]
    y32 = run_conv3x3(F_, x, w, b, split=split, f32_out=True)
    assert y32.shape == yref.shape
    assert maxerr(y32, yref) <= F32_TOL, f"fp32-accumulate check {maxerr(y32, yref)}"
    y16 = run_conv3x3(F_, x, w, b, split=split)
    assert maxerr(y16, yref) <= BF16_TOL, f"bf16 output {maxerr(y16, yref)}"
    # bf16 output must be the RNE rounding of the fp32-accumulated value (the two-source layer's bf16 form sums its two
    # reduction halves separately: an element on a rounding boundary may land on the neighbouring bf16 value)
    if split is None:
        assert np.array_equal(y16, bf16r(y32))
    else:
        d = np.abs(y16 - bf16r(y32))
        assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(np.abs(y32).max())) and float((d > 0).mean()) <= 2e-3


@pytest.mark.parametrize("act,alpha,with_res", [(1, 1.0, False), (1, 1.0, True), (0, 0.2, True), (2, 1.0, True), (2, 1.0, False)])
def test_conv3x3_epilogues(F_, act, alpha, with_res):
    rng = np.random.default_rng(5)
    x = bf16r(rng.uniform(-1, 1, (1, 64, 9, 40)))
    w = bf16r(rng.standard_normal((64, 64, 3, 3)) * 0.06)
    b = bf16r(rng.standard_normal(64) * 0.1)
    res = bf16r(rng.uniform(-1, 1, (1, 64, 9, 40))) if with_res else None
    y = run_conv3x3(F_, x, w, b, res=res, act=act, slope=0.05, alpha=alpha)
    ref = orc.conv2d(x, w, b, 1, 1, act=act, slope=0.05) * np.float32(alpha)
    if with_res:
        ref = ref + res
    assert maxerr(y, ref) <= BF16_TOL


@pytest.mark.parametrize("shape", [(1, 64, 64, 70, 300), (2, 64, 64, 33, 129), (1, 96, 64, 40, 200), (1, 16, 64, 50, 131),
                                   (3, 64, 3, 21, 67), (1, 32, 32, 1, 5), (1, 64, 64, 2, 1), (1, 48, 48, 19, 257)])
def test_conv3x3_shapes_vs_oracle(F_, shape):
    """Several strips / segments / ragged right edges / degenerate sizes."""
    n, cin, cout, h, w = shape
    rng = np.random.default_rng(h * w)
    x = bf16r(rng.uniform(-1, 1, (n, cin, h, w)))
    wt = bf16r(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5)
    b = bf16r(rng.standard_normal(cout) * 0.1)
    split = 32 if cin == 96 else None
    y = run_conv3x3(F_, x, wt, b, split=split, f32_out=True)
    ref = orc.conv2d(x, wt, b, 1, 1)
    assert maxerr(y, ref) <= F32_TOL


@pytest.mark.parametrize("shape", [(1, 40, 200), (2, 33, 129), (1, 1, 5), (1, 2, 1), (1, 70, 300), (3, 21, 67), (1, 9, 64), (1, 5, 65)])
def test_concat_layer_reduction_split_vs_oracle(F_, shape):
    """concat(32, 64) -> 64 with bf16 output runs on the reduction-split kernel (each wave of a pixel group sums half of
    the 96 input channels for all 64 outputs; the partials meet in the staging area): against the oracle at the bf16 bar
    and against the output-channel-split kernel's fp32 result (taken through its fp32 NCHW form), from which it may
    differ by the summation order only, i.e. by one bf16 step on the rare element that sits on a rounding boundary."""
    n, h, w = shape
    rng = np.random.default_rng(h * w + n)
    x = bf16r(rng.uniform(-1, 1, (n, 96, h, w)))
    wt = bf16r(rng.standard_normal((64, 96, 3, 3)) * (2.0 / (9 * 96)) ** 0.5)
    b = bf16r(rng.standard_normal(64) * 0.1)
    for act, alpha in ((1, 1.0), (0, 0.5), (2, 1.0)):
        y = run_conv3x3(F_, x, wt, b, split=32, act=act, slope=0.05, alpha=alpha)
        ref = orc.conv2d(x, wt, b, 1, 1, act=act, slope=0.05) * np.float32(alpha)
        assert maxerr(y, ref) <= BF16_TOL
        y32 = run_conv3x3(F_, x, wt, b, split=32, act=act, slope=0.05, alpha=alpha, f32_out=True)
        assert maxerr(y32, ref) <= F32_TOL
        d = np.abs(y - bf16r(y32))
        assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(np.abs(y32).max()))
        assert float((d > 0).mean()) <= 2e-3


def test_conv3x3_dgrad_form(F_, ops_golden):
    """The data gradient of a 3x3 conv is the same kernel on transposed / flipped weights."""
    z = ops_golden
    w, dy, dxref = z["convbwd.b64_64.w"], z["convbwd.b64_64.dy"], z["convbwd.b64_64.dx"]
    dyd = F_.nchw_f32_to_nhwc_bf16(dev(dy))
    wp = F_.conv3x3_pack_weights(dev(w), 64, 64, transpose_flip=True)
    dx = F_.conv3x3(dyd, wp, None, 64, out_nchw_f32=True).cpu().numpy()
    assert maxerr(dx, dxref) <= F32_TOL


def test_conv3x3_linearity(F_):
    """Size-independent property at a larger size: conv(a*x1 + x2) == a*conv(x1) + conv(x2) (no bias)."""
    rng = np.random.default_rng(9)
    shape = (1, 64, 96, 512)
    x1 = bf16r(rng.integers(-8, 8, shape) / 8.0)
    x2 = bf16r(rng.integers(-8, 8, shape) / 8.0)
    w = bf16r(rng.integers(-4, 4, (64, 64, 3, 3)) / 64.0)
    y1 = run_conv3x3(F_, x1, w, None, f32_out=True)
    y2 = run_conv3x3(F_, x2, w, None, f32_out=True)
    y12 = run_conv3x3(F_, bf16r(2.0 * x1 + x2), w, None, f32_out=True)
    # all values are small dyadic rationals: fp32 accumulation is exact
    assert np.array_equal(y12, 2.0 * y1 + y2)


def test_conv3x3_rejects_bad_arguments(F_):
    from superresolution_amd._lib import SrhipError
    x = torch.zeros((1, 4, 4, 24), dtype=torch.bfloat16, device="cuda")  # 24 channels: not a multiple of 16
    wp = torch.zeros((1024,), dtype=torch.uint8, device="cuda")
    with pytest.raises(SrhipError):
        F_.conv3x3(x, wp, None, 64)


# ------------------------------------------------------------------------------------------------
# generic direct convolution
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c48_16_1x1", "c16_16_s2p0", "c32_48_s2p1", "c48_48_3x1", "c48_48_1x3", "c48_48"])
def test_conv2d_direct_golden(F_, ops_golden, name):
    z = ops_golden
    x, w, b, yref = z[f"conv.{name}.x"], z[f"conv.{name}.w"], z[f"conv.{name}.b"], z[f"conv.{name}.y"]
    s, ph, pw = [int(v) for v in z[f"conv.{name}.cfg"]]
    xd = F_.nchw_f32_to_nhwc_bf16(dev(x))
    y = F_.conv2d_direct(xd, dev(w), dev(b), x.shape[1], stride=s, padding=(ph, pw))
    y = F_.nhwc_bf16_to_nchw_f32(y, w.shape[0]).cpu().numpy()
    assert y.shape == yref.shape
    assert maxerr(y, yref) <= BF16_TOL


# ------------------------------------------------------------------------------------------------
# memory-bound operators
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hw", [(3, 4), (48, 80), (7, 5), (1080, 1920), (2160, 3840)])
def test_make_coord_bit_exact(F_, ops_golden, hw):
    h, w = hw
    c = F_.make_coord(h, w, "cuda").cpu().numpy()
    assert np.array_equal(c[0, 0, 0, :], ops_golden[f"coord.{h}x{w}.x"])
    assert np.array_equal(c[0, 1, :, 0], ops_golden[f"coord.{h}x{w}.y"])
    assert np.array_equal(c, orc.make_coord(h, w))


@pytest.mark.parametrize("cfg", [(3, 4, 16), (6, 5, 2), (5, 7, 3), (5, 7, 4), (270, 480, 4)])
def test_rel_coord_bit_exact(F_, cfg):
    sh, sw, s = cfg
    r = F_.rel_coord(sh * s, sw * s, sh, sw, "cuda").cpu().numpy()
    assert np.array_equal(r, orc.rel_coord(sh * s, sw * s, sh, sw))


@pytest.mark.parametrize("name", ["s4", "s2", "s3"])
def test_upsample_f32_golden(F_, ops_golden, name):
    x, yref = ops_golden[f"up.{name}.x"], ops_golden[f"up.{name}.y"]
    y = F_.upsample_grid_bilinear(dev(x), yref.shape[2], yref.shape[3]).cpu().numpy()
    assert maxerr(y, yref) <= 2e-6


@pytest.mark.parametrize("name", ["s4", "s2", "s3"])
def test_upsample_nhwc_golden(F_, ops_golden, name):
    x, yref = ops_golden[f"up.{name}.x"], ops_golden[f"up.{name}.y"]
    xd = F_.nchw_f32_to_nhwc_bf16(dev(x))
    y = F_.upsample_grid_bilinear_nhwc(xd, yref.shape[2], yref.shape[3])
    y = F_.nhwc_bf16_to_nchw_f32(y, x.shape[1]).cpu().numpy()
    assert maxerr(y, yref) <= 4e-3  # one bf16 rounding of values in [-1, 1]


def test_upsample_bwd_golden(F_, ops_golden):
    z = ops_golden
    dx = F_.upsample_grid_bilinear_bwd(dev(z["upbwd.dy"]), 5, 7).cpu().numpy()
    assert maxerr(dx, z["upbwd.dx"]) <= 1e-4


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("align", [True, False])
def test_backward_warp_f32_golden(F_, ops_golden, mode, align):
    z = ops_golden
    y = F_.backward_warp(dev(z["warp.frame"]), dev(z["warp.motion"]), mode, align).cpu().numpy()
    ref = z[f"warp.{mode}.{int(align)}"]
    if mode == "nearest":
        assert np.array_equal(y, ref)
    else:
        assert maxerr(y, ref) <= 2e-6


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
def test_backward_warp_nhwc(F_, mode):
    rng = np.random.default_rng(3)
    frame = bf16r(rng.uniform(-1, 1, (2, 16, 21, 34)))
    motion = (rng.standard_normal((2, 2, 21, 34)) * 3).astype(np.float32)
    fd = F_.nchw_f32_to_nhwc_bf16(dev(frame))
    y = F_.backward_warp_nhwc(fd, dev(motion), mode, True)
    y = F_.nhwc_bf16_to_nchw_f32(y).cpu().numpy()
    ref = orc.backward_warp(frame, motion, mode, True)
    if mode == "nearest":
        assert np.array_equal(y, ref)
    else:
        assert maxerr(y, ref) <= 4e-3


def test_backward_warp_bwd_golden(F_, ops_golden):
    z = ops_golden
    dframe, dmotion = F_.backward_warp_bwd(dev(z["warp.frame"]), dev(z["warp.motion"]), dev(z["warpbwd.dy"]), True)
    assert maxerr(dframe.cpu().numpy(), z["warpbwd.dframe"]) <= 1e-5
    assert maxerr(dmotion.cpu().numpy(), z["warpbwd.dmotion"]) <= 1e-4


@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pixel_shuffle_bit_exact(F_, ops_golden, r, dtype):
    z = ops_golden
    x = dev(z[f"ps.{r}.x"] % 251).to(dtype)  # small integers: exact in bf16 too
    y = F_.pixel_shuffle(x, r)
    assert torch.equal(y.cpu(), torch.nn.functional.pixel_shuffle(x.cpu(), r))
    assert torch.equal(F_.pixel_unshuffle(y, r), x)
    if dtype == torch.float32:
        assert np.array_equal(F_.pixel_shuffle(dev(z[f"ps.{r}.x"]), r).cpu().numpy(), z[f"ps.{r}.y"])
        assert np.array_equal(F_.pixel_unshuffle(dev(z[f"pu.{r}.x"]), r).cpu().numpy(), z[f"pu.{r}.y"])


def test_pixel_shuffle_head_fused(F_):
    rng = np.random.default_rng(0)
    x = bf16r(rng.uniform(-1, 1, (2, 48, 9, 11)))
    xd = F_.nchw_f32_to_nhwc_bf16(dev(x))
    y = F_.pixel_shuffle_nhwc_to_nchw_f32(xd, 3, 4).cpu().numpy()
    assert np.array_equal(y, orc.pixel_shuffle(x, 4))


def test_pixel_shuffle_full_size_roundtrip(F_):
    """4K head: unshuffle(shuffle(x)) == x and the element multiset is preserved (checksum)."""
    x = torch.arange(48 * 540 * 960, dtype=torch.int32, device="cuda").view(1, 48, 540, 960).view(torch.float32)
    y = F_.pixel_shuffle(x, 4)
    assert y.shape == (1, 3, 2160, 3840)
    assert torch.equal(F_.pixel_unshuffle(y, 4), x)
    assert int(y.view(torch.int32).to(torch.int64).sum()) == int(x.view(torch.int32).to(torch.int64).sum())
    assert torch.equal(y[0, 1, 5::4, 2::4], x[0, 16 + 4 * 1 + 2, 1:, :])


@pytest.mark.parametrize("s", [2, 3, 4])
def test_nearest_down_bit_exact(F_, ops_golden, s):
    z = ops_golden
    y = F_.nearest_down(dev(z[f"nd.{s}.x"]), 1.0 / s).cpu().numpy()
    assert np.array_equal(y, z[f"nd.{s}.y"])


def test_concat_bit_exact(F_):
    a = torch.randn(2, 5, 7, 32, device="cuda").to(torch.bfloat16)
    b = torch.randn(2, 5, 7, 64, device="cuda").to(torch.bfloat16)
    assert torch.equal(F_.concat_nhwc(a, b), torch.cat((a, b), dim=-1))


def test_layernorm_cf_golden(F_, ops_golden):
    z = ops_golden
    y = F_.layernorm_cf(dev(z["ln.x"]), dev(z["ln.w"]), dev(z["ln.b"]), 1e-6).cpu().numpy()
    assert maxerr(y, z["ln.y"]) <= 1e-5


# ------------------------------------------------------------------------------------------------
# whole network
# ------------------------------------------------------------------------------------------------
def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 10.0 * np.log10(peak * peak / max(mse, 1e-20))


def test_supersampler_forward_golden(ss_golden):
    from superresolution_amd import SuperSampler
    z, sd = ss_golden
    m = SuperSampler(3, 10, 1).cuda().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    batch = {"lr_frames": dev(z["lr_frames"]), "hr_buffers": dev(z["hr_buffers"]), "hr_frames": dev(z["hr_frames"]),
             "scale": 4.0}
    with torch.no_grad():
        out = m(batch)
    assert "lr_frames" not in out and "hr_buffers" not in out and "hr_frames" in out
    sr = out["sr_frames"].cpu().numpy()
    ref = z["sr_frames"]
    assert sr.shape == ref.shape and sr.dtype == np.float32
    assert maxerr(sr, ref) <= BF16_TOL, maxerr(sr, ref)
    # PSNR against the HR target moves by < 0.05 dB relative to the reference output
    hr = z["hr_frames"]
    assert abs(psnr(np.clip(sr, 0, 10), hr) - psnr(np.clip(ref, 0, 10), hr)) < 0.05


def test_layernorm_cf_backward_golden_and_oracle(F_):
    """Backward of the channels-first LayerNorm against torch autograd of the reference formula
    (tests/golden/ln_bwd.npz) and the oracle; through the autograd wrapper as well.  fp32, tolerance 1e-5 relative
    to the largest gradient."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ln_bwd.npz"))
    for tag in "abc":
        x, w, b, dy = (torch.from_numpy(z[f"{tag}.{k}"]).cuda() for k in ("x", "w", "b", "dy"))
        dx, dw, db = F_.layernorm_cf_bwd(x, w, dy)
        odx, odw, odb = orc.layernorm_cf_bwd(z[f"{tag}.x"], z[f"{tag}.w"], z[f"{tag}.dy"])
        for got, gold, o in ((dx, z[f"{tag}.dx"], odx), (dw, z[f"{tag}.dw"], odw), (db, z[f"{tag}.db"], odb)):
            tol = 1e-5 * max(1.0, float(np.abs(gold).max()))
            assert np.abs(got.cpu().numpy() - gold).max() <= tol
            assert np.abs(got.cpu().numpy() - o).max() <= tol
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = F_.layernorm_cf_autograd(xr, wr, br)
        assert np.abs(y.detach().cpu().numpy() - z[f"{tag}.y"]).max() <= 5e-6
        y.backward(dy)
        assert torch.equal(xr.grad, dx) and torch.equal(wr.grad, dw) and torch.equal(br.grad, db)


@pytest.mark.parametrize("shape", [(2, 60, 33, 47), (1, 64, 16, 16), (3, 7, 5, 9), (1, 96, 12, 21)])
def test_layernorm_cf_register_form_equals_the_channel_loops(F_, shape, monkeypatch):
    """Up to 64 channels the LayerNorm kernels keep a pixel's channels in registers (one pass over x and dy) and reduce the
    parameter gradients by a butterfly over the wave: y and dx bit-equal to the channel-loop kernels (same operation
    order per pixel), dweight / dbias equal up to the order of the fp32 sums; also against torch autograd of the formula
    (CAMixerSR.py:29-37).  96 channels run the channel loops either way."""
    n, c, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(c + h)
    x = torch.randn(n, c, h, w, device="cuda", generator=g)
    wt = torch.rand(c, device="cuda", generator=g) + 0.5
    b = torch.randn(c, device="cuda", generator=g)
    dy = torch.randn(n, c, h, w, device="cuda", generator=g)
    y1 = F_.layernorm_cf(x, wt, b, 1e-6)
    g1 = F_.layernorm_cf_bwd(x, wt, dy)
    monkeypatch.setenv("SRHIP_LN_FORM", "loops")
    y0 = F_.layernorm_cf(x, wt, b, 1e-6)
    g0 = F_.layernorm_cf_bwd(x, wt, dy)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and torch.equal(g1[0], g0[0])
    for a_, b_ in zip(g1[1:], g0[1:]):
        assert float((a_ - b_).abs().max()) <= 1e-5 * max(1.0, float(b_.abs().max()))
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    u = xr.mean(1, keepdim=True)
    s = (xr - u).pow(2).mean(1, keepdim=True)
    ref = wr[:, None, None] * ((xr - u) / torch.sqrt(s + 1e-6)) + br[:, None, None]
    ref.backward(dy)
    for got, want in zip(g1, (xr.grad, wr.grad, br.grad)):
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_layernorm_cf_backward_large_is_deterministic(F_):
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(2, 60, 270, 480, device="cuda", generator=g)
    w = torch.rand(60, device="cuda", generator=g) + 0.5
    dy = torch.randn(2, 60, 270, 480, device="cuda", generator=g)
    a = F_.layernorm_cf_bwd(x, w, dy)
    b = F_.layernorm_cf_bwd(x, w, dy)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    # dbias is a plain sum: check it against a float64 reduction
    ref = dy.double().sum(dim=(0, 2, 3))
    assert (a[2].double() - ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("shape", [(2, 3, 37, 64), (1, 10, 20, 36), (1, 3, 9, 4)])
@pytest.mark.parametrize("align", [True, False])
def test_backward_warp_f32_out_of_frame_vectors(F_, shape, align):
    """10 % of the motion vectors point far out of the frame: the branch-free tap loads (clamped address, tap
    replaced by zero) must give what skipping the tap gives, i.e. the oracle's zero padding; and a repeat is
    bit-identical."""
    n, c, h, w = shape
    rng = np.random.default_rng(n * 100 + c + w)
    frame = rng.uniform(-1, 1, shape).astype(np.float32)
    motion = (rng.standard_normal((n, 2, h, w)) * 2.5).astype(np.float32)
    far = rng.random((n, 1, h, w)) < 0.1
    motion = np.where(far, motion * 1e4, motion).astype(np.float32)
    y = F_.backward_warp(dev(frame), dev(motion), "bilinear", align)
    assert torch.equal(y, F_.backward_warp(dev(frame), dev(motion), "bilinear", align))
    assert bool(torch.isfinite(y).all())
    assert maxerr(y.cpu().numpy(), orc.backward_warp(frame, motion, "bilinear", align)) <= 2e-6
