"""conv2d_auto: layers wider than one MFMA launch, stride-2 and 1x1 layers composed from 3x3 / stride-1 MFMA
launches (core/models/model_final.py:206-452 shapes) against an fp32 convolution of the same bf16-rounded
operands.  Tolerance 2e-2 (bf16 outputs; partial sums of > 64 input channels are rounded to bf16 between
launches)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # cin, cout, k, stride, pad, res, act
    (160, 64, 3, 1, 1, False, 1),
    (64, 160, 3, 1, 1, False, 1),
    (112, 64, 1, 1, 0, False, 1),
    (160, 160, 1, 1, 0, False, 1),
    (32, 48, 3, 2, 1, False, 1),
    (192, 256, 3, 2, 1, False, 1),
    (128, 64, 3, 1, 1, True, 1),
    (96, 48, 3, 1, 1, False, 0),
    (384, 128, 3, 1, 1, False, 1),
    (24, 24, 3, 1, 1, True, 1),
    (3, 24, 3, 1, 1, False, 1),
]


@pytest.mark.parametrize("cin,cout,k,stride,pad,use_res,act", CASES)
def test_conv2d_auto_matches_fp32_reference(cin, cout, k, stride, pad, use_res, act):
    from superresolution_amd import functional as F_
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    n, h, w = 2, 38, 52
    x = (torch.rand(n, cin, h, w, generator=g) - 0.3)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.5 / (cin * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    xb = x.to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(xb, wt.to(torch.bfloat16).float(), b, stride=stride, padding=pad)
    if act == 1:
        ref = torch.relu(ref)
    res = None
    if use_res:
        res = (torch.rand(ref.shape, generator=g) - 0.5).to(torch.bfloat16)
        ref = ref + res.float()
    xn = F_.nchw_f32_to_nhwc_bf16(x.cuda())
    wp, bp = torch.nn.Parameter(wt.cuda()), torch.nn.Parameter(b.cuda())
    resn = None if res is None else F_.nchw_f32_to_nhwc_bf16(res.float().cuda())
    for _ in range(2):  # second call hits the pack cache
        y = F_.conv2d_auto(xn, wp, bp, stride=stride, padding=pad, act=act, res=resn)
        assert y.shape == (n, ref.shape[2], ref.shape[3], F_.pad16(cout))
        got = y[..., :cout].permute(0, 3, 1, 2).float().cpu()
        err = (got - ref).abs().max().item()
        assert err <= 2e-2, err
        if F_.pad16(cout) != cout:
            assert float(y[..., cout:].abs().max()) == 0.0
    # the cache follows the data: an in-place weight update must be seen
    with torch.no_grad():
        wp.mul_(0.5)
    y2 = F_.conv2d_auto(xn, wp, bp, stride=stride, padding=pad, act=0)
    ref2 = torch.nn.functional.conv2d(xb, (wt * 0.5).to(torch.bfloat16).float(), b, stride=stride, padding=pad)
    assert (y2[..., :cout].permute(0, 3, 1, 2).float().cpu() - ref2).abs().max().item() <= 2e-2


def test_res_pre_mode_against_oracle():
    """dst = relu(conv + bias + res): the epilogue form that closes a chain of partial sums."""
    import numpy as np
    from oracle import oracle as orc
    from superresolution_amd import functional as F_
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 64, 20, 70, generator=g) - 0.4)
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g) * 0.1
    r = (torch.rand(1, 64, 20, 70, generator=g) - 0.5).to(torch.bfloat16).float()
    xb = x.to(torch.bfloat16).float()
    pre = orc.conv2d(xb.numpy(), wt.to(torch.bfloat16).float().numpy(), b.numpy(), 1, 1) + r.numpy()
    ref = np.maximum(pre, 0.0)
    xn = F_.nchw_f32_to_nhwc_bf16(x.cuda())
    rn = F_.nchw_f32_to_nhwc_bf16(r.cuda())
    y = F_.conv3x3(xn, F_.conv3x3_pack_weights(wt.cuda(), 64, 64), b.cuda(), 64, res=rn, act=F_.ACT_RELU, res_mode=F_.RES_PRE)
    assert np.abs(y.permute(0, 3, 1, 2).float().cpu().numpy() - ref).max() <= 2e-2


def test_channel_slices_in_place_equal_copies_bit_for_bit():
    """ca_slice / out_off (a launch reading one channel slice of a wider tensor and writing one channel slice of a
    wider tensor, with the residual read in place) against the same launches on copied slices."""
    from superresolution_amd import functional as F_
    g = torch.Generator(device="cuda").manual_seed(8)
    n, h, w = 2, 37, 150
    x = (torch.rand(n, h, w, 160, device="cuda", generator=g) - 0.4).to(torch.bfloat16)
    wt = torch.randn(48, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(48, device="cuda", generator=g) * 0.1
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    wide = (torch.rand(n, h, w, 192, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    before = wide.clone()
    # slice 64..128 of x -> channels 128..176 of `wide`, accumulated on what is already there (pre-activation)
    ref = F_.conv3x3(x[..., 64:128].contiguous(), wp, b, 48, res=before[..., 128:176].contiguous(), act=F_.ACT_RELU,
                     res_mode=F_.RES_PRE)
    F_.conv3x3(x, wp, b, 48, res=wide, act=F_.ACT_RELU, res_mode=F_.RES_PRE, ca_slice=(64, 64), out=wide, out_off=128)
    assert torch.equal(wide[..., 128:176], ref)
    assert torch.equal(wide[..., :128], before[..., :128]) and torch.equal(wide[..., 176:], before[..., 176:])
    with pytest.raises(ValueError):
        F_.conv3x3(x, wp, b, 48, ca_slice=(100, 64))
    with pytest.raises(ValueError):
        F_.conv3x3(x, wp, b, 48, out=wide, out_off=160)   # 160 + pad16(48) > 192


@pytest.mark.parametrize("chs,cout,stride", [((64, 64), 64, 1), ((192, 192), 128, 1), ((64, 32, 64), 160, 1), ((64, 16), 64, 2), ((32, 16), 48, 1)])
def test_concatenated_sources_are_read_in_place(chs, cout, stride):
    """conv2d_auto on a tuple of maps == conv2d_auto on their channel concatenation (torch.concat in front of a
    convolution, core/models/model_final.py:298,375-379,424): bit for bit when the sources fall on the same 64-channel
    slices, within the bf16 partial-sum rounding otherwise."""
    from superresolution_amd import functional as F_
    g = torch.Generator(device="cuda").manual_seed(sum(chs) + cout)
    n, h, w = 1, 22, 36
    xs = [(torch.rand(n, h, w, c, device="cuda", generator=g) - 0.4).to(torch.bfloat16) for c in chs]
    cin = sum(chs)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (1.0 / (3.0 * cin ** 0.5))
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    got = F_.conv2d_auto(tuple(xs), wt, b, stride=stride, padding=1, act=F_.ACT_RELU)
    ref = F_.conv2d_auto(torch.cat(xs, dim=-1).contiguous(), wt, b, stride=stride, padding=1, act=F_.ACT_RELU)
    assert got.shape == ref.shape
    same_slices = all(c % 64 == 0 for c in chs[:-1])
    if same_slices:
        assert torch.equal(got, ref)
    else:
        assert (got.float() - ref.float()).abs().max().item() <= 2e-2
    with pytest.raises(ValueError):
        F_.conv2d_auto((xs[0], xs[0][:, :-1]), wt, b)


@pytest.mark.parametrize("case", [
    # (x channels, x window, ci_real, dz channels, dz window, co_real)
    (112, (64, 64), 48, 160, (128, 32), 32),
    (160, (64, 64), 64, 64, (0, 64), 64),
    (48, (0, 64), 48, 96, (64, 32), 32),      # the window of x reaches past the map's last channel
    (32, (16, 16), 16, 16, (0, 16), 3),
])
def test_weight_gradient_reads_channel_windows_in_place(case):
    """srhip_conv3x3_wgrad_windows on channel windows of wider maps == srhip_conv3x3_wgrad on zero-padded copies of the
    windows, bit for bit, bias gradient included: what a window reads past its real channels only meets entries of the
    gradient that are not written."""
    from superresolution_amd import functional as F_
    cx, (xo, xt), ci_real, cz, (zo, zt), co_real = case
    g = torch.Generator().manual_seed(cx + cz)
    n, h, w = 2, 37, 150
    x = torch.relu(torch.randn(n, h, w, cx, generator=g)).to(torch.bfloat16).cuda()
    dz = (torch.randn(n, h, w, cz, generator=g) * 0.1).to(torch.bfloat16).cuda()
    w_i = ci_real + 5

    def padded(t, o, tile, real):
        out = torch.zeros(n, h, w, tile, dtype=torch.bfloat16, device="cuda")
        out[..., :real] = t[..., o:o + real]
        return out

    ref_dw = torch.full((co_real, w_i, 3, 3), 7.0, device="cuda")
    ref_db = torch.zeros(co_real, device="cuda")
    F_.conv3x3_wgrad(padded(x, xo, xt, ci_real), padded(dz, zo, zt, co_real), ref_dw, ref_db, ci_real, co_real, ci_off=5)
    got_dw = torch.full((co_real, w_i, 3, 3), 7.0, device="cuda")
    got_db = torch.zeros(co_real, device="cuda")
    F_.conv3x3_wgrad(x, dz, got_dw, got_db, ci_real, co_real, ci_off=5, x_window=(xo, xt), dz_window=(zo, zt))
    assert torch.equal(got_dw, ref_dw) and torch.equal(got_db, ref_db)
    assert float(got_dw[:, :5].min()) == 7.0 and float(got_dw[:, :5].max()) == 7.0       # outside [ci_off, ci_off + ci_real): untouched
    assert float(got_dw[:, 5:].abs().max()) > 0.0
    with pytest.raises(RuntimeError):
        F_.conv3x3_wgrad(x, dz, got_dw, got_db, ci_real, co_real, ci_off=5, x_window=(cx, xt), dz_window=(zo, zt))
