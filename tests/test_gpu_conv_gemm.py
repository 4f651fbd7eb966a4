"""GPU parity of the implicit-GEMM convolution (srhip_conv_gemm_fwd): the layers outside the 3x3 / stride 1 / <= 64-channel
shape -- core/models/model_final.py:237-246 (160 -> 160 1x1), :272-276, :329-363 (wide layers, 3x3 stride 2, concatenated
skip connections :375-379), core/arch/RepRFN.py:285-291 (ESA 1x1 and stride-2 pad-0 layers).

The whole reduction is accumulated in fp32 inside one launch, so the bar is the fp32-accumulate bar: against the CPU oracle on
the same bf16-rounded operands, |got - ref| <= 1e-3 + one bf16 rounding of the result (2^-8 |ref|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def _rand(g, *shape, scale=1.0):
    return ((torch.rand(*shape, generator=g) - 0.5) * scale).to(torch.bfloat16)


def _check(F_, srcs_real, cout, k, stride, pad, act="none", slope=0.0, alpha=1.0, with_res=False, seed=0, n=1, h=9, w=21, bias=True):
    """srcs_real: real channel counts of the concatenated maps (only the last may be padded to a multiple of 16)."""
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(seed)
    cin = sum(srcs_real)
    maps, cols = [], []
    for c in srcs_real:
        cp = F_.pad16(c)
        m = torch.zeros(n, h, w, cp, dtype=torch.bfloat16)
        m[..., :c] = _rand(g, n, h, w, c)
        maps.append(m)
        cols.append(m[..., :c])
    x_real = torch.cat(cols, dim=-1).float().permute(0, 3, 1, 2).contiguous().numpy()
    wt = (torch.randn(cout, cin, k, k, generator=g) * (1.5 / np.sqrt(cin * k * k))).to(torch.bfloat16).float()
    b = (torch.randn(cout, generator=g) * 0.1) if bias else None
    ref = orc.conv2d(x_real, wt.numpy(), None if b is None else b.numpy(), stride, pad)
    if act == "relu":
        ref = np.maximum(ref, 0.0)
    elif act == "lrelu":
        ref = np.where(ref > 0, ref, ref * slope)
    ref = ref * alpha
    ho, wo = ref.shape[-2:]
    res = None
    if with_res:
        res = torch.zeros(n, ho, wo, F_.pad16(cout), dtype=torch.bfloat16)
        res[..., :cout] = _rand(g, n, ho, wo, cout)
        ref = ref + res[..., :cout].float().permute(0, 3, 1, 2).numpy()
    cin_pad = sum(m.shape[-1] for m in maps)
    # weight columns follow the padded concatenation: only the last source may carry padding
    wp = F_.conv_gemm_pack_weights(wt.cuda(), cin_pad)
    code = {"none": F_.ACT_NONE, "relu": F_.ACT_RELU, "lrelu": F_.ACT_LRELU}[act]
    out = F_.conv_gemm(tuple(m.cuda() for m in maps) if len(maps) > 1 else maps[0].cuda(), wp, None if b is None else b.cuda(), cout, k,
                       stride=stride, padding=pad, act=code, slope=slope, alpha=alpha, res=None if res is None else res.cuda())
    torch.cuda.synchronize()
    assert tuple(out.shape) == (n, ho, wo, F_.pad16(cout))
    got = out.float().cpu()
    assert float(got[..., cout:].abs().max()) == 0.0 if F_.pad16(cout) > cout else True
    got = got[..., :cout].permute(0, 3, 1, 2).numpy()
    tol = 1e-3 + 2.0 ** -8 * np.abs(ref)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - ref).max()))


@pytest.mark.parametrize("case", [
    dict(srcs_real=[160], cout=160, k=1, stride=1, pad=0, act="relu", h=20, w=37),            # G-buffer encoder 1x1
    dict(srcs_real=[192, 128], cout=256, k=3, stride=1, pad=1, act="relu"),                   # decoder layer on a skip concat
    dict(srcs_real=[64, 32, 64], cout=128, k=3, stride=1, pad=1, act="relu", n=2, w=150),     # three maps, two strips, batch 2
    dict(srcs_real=[64], cout=128, k=3, stride=2, pad=1, act="relu", h=22, w=40),             # down-sampling, even size
    dict(srcs_real=[128], cout=192, k=3, stride=2, pad=1, act="relu", h=21, w=39),            # ... odd size
    dict(srcs_real=[24], cout=64, k=1, stride=1, pad=0, act="none"),                          # padded input channels
    dict(srcs_real=[64], cout=1, k=3, stride=1, pad=1, act="relu"),                           # one output channel
    dict(srcs_real=[64], cout=24, k=3, stride=1, pad=1, act="relu", w=300),                   # 24 outputs, three strips
    dict(srcs_real=[16], cout=16, k=3, stride=2, pad=0, act="none", h=25, w=31),              # ESA stride-2 pad-0
    dict(srcs_real=[48], cout=16, k=1, stride=1, pad=0, act="none", bias=False),              # ESA 1x1, no bias
    dict(srcs_real=[256], cout=256, k=3, stride=1, pad=1, act="lrelu", slope=0.05, alpha=0.5, with_res=True, h=5, w=7),
    dict(srcs_real=[112], cout=112, k=3, stride=1, pad=1, act="relu", h=3, w=130),            # 7 x 16 channels
    dict(srcs_real=[64], cout=64, k=1, stride=2, pad=0, act="none", h=10, w=18),              # 1x1 stride 2
    dict(srcs_real=[384], cout=64, k=3, stride=1, pad=1, act="relu", h=4, w=5, with_res=True),
])
def test_conv_gemm_vs_oracle(F_, case):
    _check(F_, seed=len(str(case)), **case)


def test_conv_gemm_channel_slice_source_and_wider_destination(F_):
    """A source may be a channel window of a wider tensor, and the result may land in the first channels of a wider one."""
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(5)
    big = _rand(g, 1, 8, 19, 96).cuda()
    wt = (torch.randn(32, 32, 3, 3, generator=g) * 0.08).to(torch.bfloat16).float()
    wp = F_.conv_gemm_pack_weights(wt.cuda(), 32)
    dst = torch.full((1, 8, 19, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    F_.conv_gemm((big, 40, 32), wp, None, 32, 3, stride=1, padding=1, out=dst)
    ref = orc.conv2d(big[..., 40:72].float().permute(0, 3, 1, 2).contiguous().cpu().numpy(), wt.numpy(), None, 1, 1)
    got = dst[..., :32].float().permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-3 + 2.0 ** -8 * np.abs(ref).max()
    assert torch.equal(dst[..., 32:], torch.full((1, 8, 19, 32), 7.0, dtype=torch.bfloat16, device="cuda"))


def test_conv_gemm_rejects_bad_arguments(F_):
    from superresolution_amd._lib import SrhipError
    x = torch.zeros(1, 8, 8, 32, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        F_.conv_gemm_pack_weights(torch.zeros(8, 32, 5, 5, device="cuda"), 32)
    wp = F_.conv_gemm_pack_weights(torch.zeros(8, 32, 3, 3, device="cuda"), 32)
    with pytest.raises(ValueError):
        F_.conv_gemm((x, 4, 16), wp, None, 8, 3, padding=1)            # window offset not a multiple of 8
    with pytest.raises(SrhipError):
        F_.conv_gemm(x, wp, None, 8, 3, stride=3, padding=1)
    with pytest.raises(SrhipError):
        F_.conv_gemm(x, wp, None, 8, 3, padding=1, alpha=0.0)



@pytest.mark.parametrize("case", [(160, 128, 3, 9, 21), (64, 64, 1, 7, 130), (384, 48, 3, 4, 5), (96, 24, 3, 6, 40)])
def test_conv_gemm_mask_epilogue_equals_a_mask_pass(F_, case):
    """res_mode = mask: the result is zeroed where `res` (the ReLU output in front of a data-gradient launch,
    core/trainer.py:160 through nn.ReLU) is not positive -- bit-equal to the launch followed by relu_mask."""
    cin, cout, k, h, w = case
    g = torch.Generator().manual_seed(cin + cout)
    x = _rand(g, 2, h, w, cin).cuda()
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    wp = F_.conv_gemm_pack_weights(wt, cin)
    y = torch.relu(_rand(g, 2, h, w, F_.pad16(cout))).cuda()
    got = F_.conv_gemm(x, wp, None, cout, k, stride=1, padding=k // 2, res=y, res_mode=F_.RES_MASK, alpha=0.5)
    ref = F_.relu_mask(F_.conv_gemm(x, wp, None, cout, k, stride=1, padding=k // 2, alpha=0.5), y)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert bool((got[y <= 0] == 0).all())


@pytest.mark.parametrize("case", [(160, 160, 9, 200, 1), (64, 40, 7, 130, 2), (96, 128, 6, 21, 1), (48, 24, 5, 300, 1)])
def test_conv_gemm_four_row_tiles_equal_two_row_tiles(F_, case, monkeypatch):
    """The four-row work-group tile (large images with 5 output tiles by default) runs the same reduction in the same order
    as the two-row tile: bit-equal results, including rows past the image's last (h not a multiple of 4)."""
    cin, cout, h, w, n = case
    g = torch.Generator().manual_seed(cin * cout)
    x = _rand(g, n, h, w, cin).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    wp = F_.conv_gemm_pack_weights(wt, cin)
    res = _rand(g, n, h, w, F_.pad16(cout)).cuda()
    outs = []
    for force in ("0", "1"):
        monkeypatch.setenv("SRHIP_GEMM_ROWS4", force)
        outs.append(F_.conv_gemm(x, wp, b, cout, 3, stride=1, padding=1, act=F_.ACT_LRELU, slope=0.1, alpha=0.5, res=res))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
