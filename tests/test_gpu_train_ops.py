"""GPU parity tests of the backward-pass operators (weight gradient, masked data gradient, the
gather form of the up-sampling backward, element-wise helpers), through the C ABI, against the CPU
oracle and the reference-generated goldens.

Tolerances: fp32 outputs of bf16-representable operands <= 1e-3 relative to the largest magnitude
(fp32 accumulate); bf16 outputs <= 2e-2."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16r(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run_wgrad(F_, x, dy, w_shape, ci_off=0, dw=None, accumulate=False):
    co, ci_total = w_shape[:2]
    xa = F_.nchw_f32_to_nhwc_bf16(dev(x))
    dz = F_.nchw_f32_to_nhwc_bf16(dev(dy))
    if dw is None:
        dw = torch.full(w_shape, 7.0, dtype=torch.float32, device="cuda")  # must be overwritten
    db = torch.full((co,), 7.0, dtype=torch.float32, device="cuda")
    F_.conv3x3_wgrad(xa, dz, dw, db, x.shape[1], co, ci_off=ci_off, accumulate=accumulate)
    torch.cuda.synchronize()
    return dw, db


@pytest.mark.parametrize("name", ["b64_64", "b10_64", "b64_3"])
def test_wgrad_golden(F_, ops_golden, name):
    z = ops_golden
    x, dy = bf16r(z[f"convbwd.{name}.x"]), bf16r(z[f"convbwd.{name}.dy"])
    dw_ref, db_ref = orc.conv2d_bwd_weight(x, dy, z[f"convbwd.{name}.w"].shape, padding=1)
    # the oracle on the unrounded operands is pinned to the reference's gradients
    dw_g, db_g = orc.conv2d_bwd_weight(z[f"convbwd.{name}.x"], z[f"convbwd.{name}.dy"], z[f"convbwd.{name}.w"].shape, padding=1)
    assert relerr(dw_g, z[f"convbwd.{name}.dw"]) <= 1e-4 and relerr(db_g, z[f"convbwd.{name}.db"]) <= 1e-4
    dw, db = run_wgrad(F_, x, dy, dw_ref.shape)
    assert relerr(dw.cpu().numpy(), dw_ref) <= 1e-3
    assert relerr(db.cpu().numpy(), db_ref) <= 1e-3
    # and within bf16 operand rounding of the reference's fp32 gradient
    assert relerr(dw.cpu().numpy(), z[f"convbwd.{name}.dw"]) <= 2e-2


@pytest.mark.parametrize("shape", [
    (1, 64, 64, 9, 130),     # strip edge + 2 columns
    (2, 32, 32, 33, 257),    # one tile, pixel-split over 4 waves
    (1, 16, 64, 5, 128),     # padded input channels
    (1, 3, 32, 17, 40),      # first LR layer
    (1, 64, 16, 12, 300),    # padded output channels
    (3, 32, 64, 140, 70),    # several segments / images
    (1, 64, 32, 1, 1),       # single pixel
])
def test_wgrad_shapes_vs_oracle(F_, shape):
    n, ci, co, h, w = shape
    rng = np.random.default_rng(ci * 1000 + co + h)
    x = bf16r(rng.standard_normal((n, ci, h, w)))
    dy = bf16r(rng.standard_normal((n, co, h, w)))
    dw_ref, db_ref = orc.conv2d_bwd_weight(x, dy, (co, ci, 3, 3), padding=1)
    dw, db = run_wgrad(F_, x, dy, (co, ci, 3, 3))
    assert relerr(dw.cpu().numpy(), dw_ref) <= 1e-3
    assert relerr(db.cpu().numpy(), db_ref) <= 1e-3


def test_wgrad_concat_window_and_accumulate(F_):
    """A concat(32, 64) layer takes one call per source, each filling its input-channel window."""
    rng = np.random.default_rng(5)
    x = bf16r(rng.standard_normal((1, 96, 20, 150)))
    dy = bf16r(rng.standard_normal((1, 64, 20, 150)))
    dw_ref, db_ref = orc.conv2d_bwd_weight(x, dy, (64, 96, 3, 3), padding=1)
    dw, _ = run_wgrad(F_, x[:, :32], dy, (64, 96, 3, 3), ci_off=0)
    dw, db = run_wgrad(F_, x[:, 32:], dy, (64, 96, 3, 3), ci_off=32, dw=dw)
    assert relerr(dw.cpu().numpy(), dw_ref) <= 1e-3
    # accumulate: a second pass doubles the window and the bias gradient
    xa = F_.nchw_f32_to_nhwc_bf16(dev(x[:, 32:]))
    dz = F_.nchw_f32_to_nhwc_bf16(dev(dy))
    F_.conv3x3_wgrad(xa, dz, dw, db, 64, 64, ci_off=32, accumulate=True)
    torch.cuda.synchronize()
    ref2 = dw_ref.copy()
    ref2[:, 32:] *= 2
    assert relerr(dw.cpu().numpy(), ref2) <= 1e-3
    assert relerr(db.cpu().numpy(), 2 * db_ref) <= 1e-3


@pytest.mark.parametrize("case", [((96, 64), 112, 2, 21, 40), ((160,), 256, 1, 9, 130), ((384,), 256, 2, 8, 12), ((48,), 24, 1, 5, 7)])
def test_wgrad_of_a_wide_layer_in_batched_jobs_vs_oracle(F_, case, monkeypatch):
    """The channel-slice pairs of a layer with more than 64 input / output channels (core/models/model_final.py:329-379)
    go out as jobs of one launch per tile shape; against the CPU oracle, against one launch per pair, and repeatable."""
    from superresolution_amd import autograd_ops as A_
    srcs_c, cout, n, h, w = case
    cin = sum(srcs_c)
    rng = np.random.default_rng(cin + cout)
    x = bf16r(rng.standard_normal((n, cin, h, w)))
    dy = bf16r(rng.standard_normal((n, cout, h, w)))
    dw_ref, db_ref = orc.conv2d_bwd_weight(x, dy, (cout, cin, 3, 3), padding=1)
    maps, k0 = [], 0
    for c in srcs_c:
        cp = F_.pad16(c)
        m = torch.zeros(n, h, w, cp, dtype=torch.bfloat16)
        m[..., :c] = torch.from_numpy(x[:, k0:k0 + c]).permute(0, 2, 3, 1).to(torch.bfloat16)
        maps.append(m.cuda())
        k0 += c
    # only the last source may carry padding for the concatenated channel index to stay contiguous
    dz = torch.zeros(n, h, w, F_.pad16(cout), dtype=torch.bfloat16)
    dz[..., :cout] = torch.from_numpy(dy).permute(0, 2, 3, 1).to(torch.bfloat16)
    dz = dz.cuda()
    wt = torch.zeros(cout, cin, 3, 3, device="cuda")
    dw, db = A_._auto_wgrad(tuple(maps), dz, wt)
    dw2, db2 = A_._auto_wgrad(tuple(maps), dz, wt)
    monkeypatch.setenv("SRHIP_NO_WGRAD_MULTI", "1")
    dw1, db1 = A_._auto_wgrad(tuple(maps), dz, wt)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert relerr(dw.cpu().numpy(), dw_ref) <= 1e-3 and relerr(db.cpu().numpy(), db_ref) <= 1e-3
    assert relerr(dw.cpu().numpy(), dw1.cpu().numpy()) <= 1e-5 and relerr(db.cpu().numpy(), db1.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("case", [((160,), 160, 2, 9, 70), ((112,), 64, 1, 20, 33), ((64,), 16, 1, 7, 130), ((32, 64), 48, 1, 5, 9)])
def test_wgrad_of_1x1_layers_on_the_centre_tap_form(F_, case, monkeypatch):
    """1x1 layers (core/models/model_final.py:237-246): the weight-gradient kernel's one-tap form against the CPU oracle and
    against the centre tap of the full 3x3 result."""
    from superresolution_amd import autograd_ops as A_
    srcs_c, cout, n, h, w = case
    cin = sum(srcs_c)
    rng = np.random.default_rng(cin * 3 + cout)
    x = bf16r(rng.standard_normal((n, cin, h, w)))
    dy = bf16r(rng.standard_normal((n, cout, h, w)))
    dw_ref, db_ref = orc.conv2d_bwd_weight(x, dy, (cout, cin, 1, 1), padding=0)
    maps, k0 = [], 0
    for c in srcs_c:
        m = torch.zeros(n, h, w, F_.pad16(c), dtype=torch.bfloat16)
        m[..., :c] = torch.from_numpy(x[:, k0:k0 + c]).permute(0, 2, 3, 1).to(torch.bfloat16)
        maps.append(m.cuda())
        k0 += c
    dz = torch.zeros(n, h, w, F_.pad16(cout), dtype=torch.bfloat16)
    dz[..., :cout] = torch.from_numpy(dy).permute(0, 2, 3, 1).to(torch.bfloat16)
    dz = dz.cuda()
    wt = torch.zeros(cout, cin, 1, 1, device="cuda")
    dw, db = A_._auto_wgrad(tuple(maps), dz, wt)
    monkeypatch.setenv("SRHIP_NO_WGRAD_1X1", "1")
    dw3, db3 = A_._auto_wgrad(tuple(maps), dz, wt)
    torch.cuda.synchronize()
    assert tuple(dw.shape) == (cout, cin, 1, 1) and tuple(dw3.shape) == (cout, cin, 1, 1)
    assert relerr(dw.cpu().numpy(), dw_ref) <= 1e-3 and relerr(db.cpu().numpy(), db_ref) <= 1e-3
    assert relerr(dw.cpu().numpy(), dw3.cpu().numpy()) <= 1e-5 and relerr(db.cpu().numpy(), db3.cpu().numpy()) <= 1e-5


def test_wgrad_is_deterministic(F_):
    rng = np.random.default_rng(6)
    x = bf16r(rng.standard_normal((2, 64, 150, 260)))
    dy = bf16r(rng.standard_normal((2, 64, 150, 260)))
    a, _ = run_wgrad(F_, x, dy, (64, 64, 3, 3))
    b, _ = run_wgrad(F_, x, dy, (64, 64, 3, 3))
    assert torch.equal(a, b)


def test_wgrad_rejects_bad_arguments(F_):
    from superresolution_amd._lib import SrhipError
    x = torch.zeros((1, 4, 4, 48), dtype=torch.bfloat16, device="cuda")
    dz = torch.zeros((1, 4, 4, 64), dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros((64, 48, 3, 3), device="cuda")
    with pytest.raises(SrhipError):
        F_.conv3x3_wgrad(x, dz, dw, None, 48, 64)  # 48 padded channels are not a wgrad tile shape
    x = torch.zeros((1, 4, 4, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SrhipError):
        F_.conv3x3_wgrad(x, dz, dw, None, 64, 64)  # window [0,64) outside the weight's 48 inputs


def test_conv3x3_mask_mode_is_relu_backward(F_):
    """dgrad with the MASK epilogue = relu'(y) * conv_transpose(dz): compare with the oracle's
    input gradient masked by the saved activation."""
    rng = np.random.default_rng(7)
    w = bf16r(rng.standard_normal((64, 64, 3, 3)) * 0.05)
    dz = bf16r(rng.standard_normal((1, 64, 21, 140)))
    y = bf16r(np.maximum(rng.standard_normal((1, 64, 21, 140)), 0))  # saved ReLU output (half zeros)
    dx_ref = orc.conv2d_bwd_input(dz, w, (1, 64, 21, 140), padding=1) * (y > 0) * 0.2
    wp = F_.conv3x3_pack_weights(dev(w), 64, 64, transpose_flip=True)
    out = F_.conv3x3(F_.nchw_f32_to_nhwc_bf16(dev(dz)), wp, None, 64, res=F_.nchw_f32_to_nhwc_bf16(dev(y)),
                     alpha=0.2, res_mode=F_.RES_MASK)
    got = F_.nhwc_bf16_to_nchw_f32(out, 64).cpu().numpy()
    assert np.array_equal(got == 0, dx_ref == 0) or float(np.mean((got == 0) != (dx_ref == 0))) < 1e-4
    assert np.abs(got - dx_ref).max() <= 2e-2


def test_relu_mask_and_add(F_):
    rng = np.random.default_rng(8)
    a = torch.from_numpy(rng.standard_normal((2, 9, 11, 64)).astype(np.float32)).to(torch.bfloat16).cuda()
    b = torch.from_numpy(rng.standard_normal((2, 9, 11, 64)).astype(np.float32)).to(torch.bfloat16).cuda()
    m = F_.relu_mask(a, b)
    assert torch.equal(m, torch.where(b > 0, a, torch.zeros_like(a)))
    s = F_.add_bf16(a, b)
    assert torch.equal(s, (a.float() + b.float()).to(torch.bfloat16))
    F_.relu_mask(a, b, out=a)  # in place
    assert torch.equal(a, m)


@pytest.mark.parametrize("cfg", [(5, 7, 20, 28), (6, 5, 12, 10), (4, 4, 9, 10), (8, 8, 8, 8)])
def test_upsample_bwd_nhwc_vs_oracle(F_, cfg):
    sh, sw, th, tw = cfg
    rng = np.random.default_rng(sh * 100 + tw)
    dy = bf16r(rng.standard_normal((2, 32, th, tw)))
    x = np.zeros((2, 32, sh, sw), np.float32)
    grid = np.broadcast_to(np.transpose(orc.make_coord(th, tw), (0, 2, 3, 1)), (2, th, tw, 2)).copy()
    dx_ref, _ = orc.grid_sample_bwd(x, grid, dy, align_corners=False, need_dgrid=False)
    got = F_.upsample_grid_bilinear_bwd_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(dy)), sh, sw)
    got = F_.nhwc_bf16_to_nchw_f32(got, 32).cpu().numpy()
    assert relerr(got, dx_ref) <= 8e-3  # one bf16 rounding of the fp32 sum
    # fused ReLU mask
    y = bf16r(np.maximum(rng.standard_normal((2, 32, sh, sw)), 0))
    got_m = F_.upsample_grid_bilinear_bwd_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(dy)), sh, sw, relu_out=F_.nchw_f32_to_nhwc_bf16(dev(y)))
    got_m = F_.nhwc_bf16_to_nchw_f32(got_m, 32).cpu().numpy()
    assert np.array_equal(got_m, got * (y > 0))


def test_upsample_bwd_nhwc_golden(F_, ops_golden):
    z = ops_golden
    dy = bf16r(z["upbwd.dy"])
    got = F_.upsample_grid_bilinear_bwd_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(dy)), 5, 7)
    got = F_.nhwc_bf16_to_nchw_f32(got, 32).cpu().numpy()
    assert relerr(got, z["upbwd.dx"]) <= 2e-2


@pytest.mark.parametrize("cfg", [(1, 1), (2, 3), (5, 7), (33, 20), (64, 64)])
def test_upsample_bwd_integer_ratio_form_equals_the_general_gather(F_, cfg, monkeypatch):
    """x4 (the network's ratio, core/models/model.py:153,162) runs a form with 8 + 8 tap weights per thread instead of
    one per visited tap: same products, same order -- bit-equal, borders included."""
    sh, sw = cfg
    g = torch.Generator().manual_seed(sh * 50 + sw)
    dy = (torch.rand(2, 4 * sh, 4 * sw, 32, generator=g) - 0.5).to(torch.bfloat16).cuda()
    y = torch.relu(torch.rand(2, sh, sw, 32, generator=g) - 0.4).to(torch.bfloat16).cuda()
    fast = F_.upsample_grid_bilinear_bwd_nhwc(dy, sh, sw, relu_out=y)
    fast_nomask = F_.upsample_grid_bilinear_bwd_nhwc(dy, sh, sw)
    monkeypatch.setenv("SRHIP_NO_UPBWD_INT", "1")
    slow = F_.upsample_grid_bilinear_bwd_nhwc(dy, sh, sw, relu_out=y)
    slow_nomask = F_.upsample_grid_bilinear_bwd_nhwc(dy, sh, sw)
    torch.cuda.synchronize()
    assert torch.equal(fast, slow) and torch.equal(fast_nomask, slow_nomask)
