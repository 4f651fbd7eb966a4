"""GPU parity of the fused residual-pair kernel (srhip_conv3x3_pair_fwd): SRResidualBlock (core/models/model.py:22-33)
and Residual(ConvActivation, ConvActivation) (core/models/model.py:17-19,70-73) as ONE launch.

Checked against (a) the CPU oracle's composition of the two layers with the intermediate map rounded to bf16 exactly
where the kernels round it (tolerance 2e-2 on bf16 outputs, as for every convolution; the fp32-accumulate bar of 1e-3
is checked on the intermediate map, which the kernel can write out), and (b) the two single-layer launches it replaces
(the second layer is the same arithmetic; the first adds the bias in front of the products instead of behind them, so a
rare element of the intermediate map may land on the neighbouring bf16 value)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _case(seed, n, h, w, scale=0.06):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16)
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * scale).to(torch.bfloat16).float()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) * scale).to(torch.bfloat16).float()
    b1 = (torch.randn(64, generator=g) * 0.1).to(torch.bfloat16).float()
    b2 = (torch.randn(64, generator=g) * 0.1).to(torch.bfloat16).float()
    return x, w1, b1, w2, b2


def _oracle_pair(x, w1, b1, w2, b2, relu2, alpha):
    """fp32 composition on the CPU oracle with the intermediate map rounded to bf16 (NCHW arrays)."""
    from oracle import oracle as orc
    xn = x.float().permute(0, 3, 1, 2).contiguous().numpy()
    mid32 = np.maximum(orc.conv2d(xn, w1.numpy(), b1.numpy(), 1, 1), 0.0)
    mid = _bf16_round(mid32)
    t = orc.conv2d(mid, w2.numpy(), b2.numpy(), 1, 1)
    if relu2:
        t = np.maximum(t, 0.0)
    return mid32, alpha * t + xn


@pytest.mark.parametrize("shape", [(1, 20, 70), (2, 37, 130), (1, 64, 62), (1, 5, 3), (1, 3, 63), (1, 9, 125), (3, 1, 1)])
@pytest.mark.parametrize("form", ["sr_block", "residual"])
def test_pair_matches_oracle_and_two_launches(F_, shape, form):
    n, h, w = shape
    relu2, alpha = (False, 0.2) if form == "sr_block" else (True, 1.0)
    x, w1, b1, w2, b2 = _case(100 * h + w + n, n, h, w)
    xc = x.cuda()
    wp1 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64)
    wp2 = F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    act2 = F_.ACT_RELU if relu2 else F_.ACT_NONE
    out, mid = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=act2, alpha=alpha, keep_mid=True)
    out_nomid = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=act2, alpha=alpha)
    torch.cuda.synchronize()
    assert torch.equal(out, out_nomid), "the form that also writes the intermediate map must give the same output"
    # (b) the two launches the kernel replaces
    t = F_.conv3x3(xc, wp1, b1.cuda(), 64, act=F_.ACT_RELU)
    ref2 = F_.conv3x3(t, wp2, b2.cuda(), 64, res=xc, act=act2, alpha=alpha)
    dm = (mid.float() - t.float()).abs()
    assert float(dm.max()) <= 2.0 ** -7 * max(1.0, float(t.float().abs().max())), "intermediate map differs by more than one bf16 step"
    assert float((dm > 0).float().mean()) <= 0.02
    assert float((out.float() - ref2.float()).abs().max()) <= 1e-2
    # (a) the oracle: intermediate map at the fp32-accumulate bar, output at the bf16 bar
    mid32, ref = _oracle_pair(x, w1, b1, w2, b2, relu2, alpha)
    got_mid = mid.float().permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got_mid - mid32).max() <= 1e-3 + 2.0 ** -8 * np.abs(mid32).max()   # fp32 accumulate, then ONE bf16 rounding
    got = out.float().permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-2


def test_pair_second_layer_is_bit_equal_given_the_same_intermediate(F_):
    """With the kernel's own intermediate map fed to the single-layer kernel, the outputs agree bit for bit: the second
    stage is the same arithmetic in the same order."""
    x, w1, b1, w2, b2 = _case(7, 2, 45, 200)
    xc = x.cuda()
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    for act2, alpha in ((F_.ACT_NONE, 0.2), (F_.ACT_RELU, 1.0), (F_.ACT_LRELU, 0.5)):
        out, mid = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=act2, slope2=0.05, alpha=alpha, keep_mid=True)
        ref = F_.conv3x3(mid, wp2, b2.cuda(), 64, res=xc, act=act2, slope=0.05, alpha=alpha)
        assert torch.equal(out, ref)


def test_pair_zero_padding_of_the_intermediate_map(F_):
    """The second convolution must see ZEROS outside the image, not the first layer evaluated on padding: with a large
    first-layer bias the two differ by far more than the tolerance along every border."""
    n, h, w = 1, 70, 190     # several strips and, with few rows per item, several segments
    x, w1, b1, w2, b2 = _case(11, n, h, w)
    b1 = b1 + 2.0
    xc = x.cuda()
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    out = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2)
    _, ref = _oracle_pair(x, w1, b1, w2, b2, False, 0.2)
    got = out.float().permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - ref).max() <= 4e-2    # values are ~10x larger here (bias 2.0): bf16 step of the output


def test_pair_rejects_bad_arguments(F_):
    from superresolution_amd._lib import SrhipError
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device="cuda")
    wp = F_.conv3x3_pack_weights(torch.zeros(64, 64, 3, 3, device="cuda"), 64, 64)
    with pytest.raises(ValueError):
        F_.conv3x3_pair(x[..., :32].contiguous(), wp, None, wp, None)
    with pytest.raises(ValueError):
        F_.conv3x3_pair(x, wp, None, wp, None, out=x)
    with pytest.raises(SrhipError):
        F_.conv3x3_pair(x, wp, None, wp, None, alpha=0.0)


@pytest.mark.parametrize("shape", [(1, 20, 70), (2, 37, 130), (1, 5, 3), (1, 9, 125), (3, 1, 1), (1, 7, 92), (1, 6, 93), (2, 11, 185),
                                   (1, 40, 300)])
@pytest.mark.parametrize("cout2", [1, 3, 5, 8, 32])
def test_pair_head_matches_oracle_and_two_launches(F_, shape, cout2):
    """Head form: conv 64 -> 64 + ReLU -> conv 64 -> cout2 as fp32 NCHW in one launch (core/models/model.py:111-115).
    Up to 8 output channels without the intermediate map stored run as three first-layer waves plus one wave that
    sums the kernel rows in registers (92-column strips: widths around 92 / 184 meet its strip borders); the other
    cases keep two waves per layer.  The two differ in the fp32 summation order only."""
    from oracle import oracle as orc
    n, h, w = shape
    x, w1, b1, w2, b2 = _case(300 * h + w + cout2, n, h, w)
    w2, b2 = w2[:cout2].contiguous(), b2[:cout2].contiguous()
    xc = x.cuda()
    wp1 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64)
    wp2 = F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    out, mid = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), cout2, keep_mid=True)
    out_nomid = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), cout2)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (n, cout2, h, w) and out.dtype == torch.float32
    assert bool(torch.isfinite(out_nomid).all())
    assert float((out - out_nomid).abs().max()) <= 1e-4
    assert torch.equal(out_nomid, F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), cout2)), "not repeatable"
    # the two launches it replaces, on the kernel's own intermediate map: same arithmetic up to the summation order
    ref2 = F_.conv3x3(mid, wp2, b2.cuda(), cout2, act=F_.ACT_NONE, out_nchw_f32=True)
    assert float((out - ref2).abs().max()) <= 1e-4
    t = F_.conv3x3(xc, wp1, b1.cuda(), 64, act=F_.ACT_RELU)
    assert float((mid.float() - t.float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(t.float().abs().max()))
    # the oracle with the intermediate map rounded where the kernel rounds it: fp32 accumulate, fp32 output
    xn = x.float().permute(0, 3, 1, 2).contiguous().numpy()
    mid32 = np.maximum(orc.conv2d(xn, w1.numpy(), b1.numpy(), 1, 1), 0.0)
    ref = orc.conv2d(_bf16_round(mid32), w2.numpy(), b2.numpy(), 1, 1)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-2      # a few intermediate elements land on the neighbouring bf16 value
    ref_own = orc.conv2d(mid.float().permute(0, 3, 1, 2).cpu().numpy(), w2.numpy(), b2.numpy(), 1, 1)
    assert np.abs(out.cpu().numpy() - ref_own).max() <= 1e-3  # the fp32-accumulate bar on the kernel's own intermediate map
    assert np.abs(out_nomid.cpu().numpy() - ref_own).max() <= 1e-3


@pytest.mark.parametrize("act2,alpha", [(1, 1.0), (2, 0.5), (0, 0.2)])
def test_pair_head_activation_scale_and_zero_padding(F_, act2, alpha):
    """The narrow layer's epilogue (activation, alpha) and the zero padding of the intermediate map on every border and
    at every segment / strip seam: a large first-layer bias makes 'first layer evaluated on padding' differ from zeros
    by far more than the tolerance."""
    from oracle import oracle as orc
    n, h, w, cout2 = 1, 70, 190, 3
    x, w1, b1, w2, b2 = _case(17 + act2, n, h, w)
    b1 = b1 + 2.0
    w2, b2 = w2[:cout2].contiguous(), b2[:cout2].contiguous()
    xc = x.cuda()
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    out, mid = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), cout2, act2=act2, slope2=0.05, alpha=alpha, keep_mid=True)
    out3 = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), cout2, act2=act2, slope2=0.05, alpha=alpha)
    ref = orc.conv2d(mid.float().permute(0, 3, 1, 2).cpu().numpy(), w2.numpy(), b2.numpy(), 1, 1, act=act2, slope=0.05) * np.float32(alpha)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-3      # values ~10x larger (bias 2.0)
    assert np.abs(out3.cpu().numpy() - ref).max() <= 2e-3


def test_pair_head_full_size_partitions(F_):
    """4K and a many-image shape: the two-class and range partitions of the 92-column strips; the three-wave form
    against the two-plus-two form (which stores the intermediate map) on device-generated data."""
    _, w1, b1, w2, b2 = _case(5, 1, 8, 8)
    w2, b2 = w2[:3].contiguous(), b2[:3].contiguous()
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    for n, h, w in ((1, 2160, 3840), (16, 512, 1024), (9, 600, 1000)):
        g = torch.Generator(device="cuda").manual_seed(h + n)
        xc = (torch.rand(n, h, w, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        out3 = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), 3)
        out2, _ = F_.conv3x3_pair_head(xc, wp1, b1.cuda(), wp2, b2.cuda(), 3, keep_mid=True)
        assert bool(torch.isfinite(out3).all())
        assert float((out3 - out2).abs().max()) <= 1e-4, (n, h, w)


def test_pair_head_rejects_bad_arguments(F_):
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device="cuda")
    wp = F_.conv3x3_pack_weights(torch.zeros(64, 64, 3, 3, device="cuda"), 64, 64)
    with pytest.raises(ValueError):
        F_.conv3x3_pair_head(x, wp, None, wp, None, 40)
    with pytest.raises(ValueError):
        F_.conv3x3_pair_head(x[..., :32].contiguous(), wp, None, wp, None, 3)


@pytest.mark.parametrize("shape,kind", [((1, 1080, 1920), "two-class"), ((9, 600, 1000), "two-class"), ((16, 512, 1024), "range"),
                                        ((2, 2160, 3840), "uniform")])
def test_pair_work_partitions(F_, shape, kind):
    """The launch cuts the strips into work items in three ways (uniform segments; shorter main segments plus the rows
    left at the bottom of every strip for the workgroups a uniform grid leaves idle; equal row ranges of the strip-major
    row sequence when the strips outnumber the workgroups unevenly).  Whatever the cut, every output row is produced
    once, from the same inputs: given the kernel's own intermediate map the single-layer kernel reproduces the output
    bit for bit, the intermediate map equals the single-layer first layer up to the bias-order bf16 step, and the form
    that does not store the intermediate map gives the same output."""
    n, h, w = shape
    x, w1, b1, w2, b2 = _case(h + w + n, 1, 8, 8)
    g = torch.Generator(device="cuda").manual_seed(h * 7 + n)
    xc = (torch.rand(n, h, w, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    out, mid = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2, keep_mid=True)
    assert torch.equal(out, F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2)), kind
    assert torch.equal(out, F_.conv3x3(mid, wp2, b2.cuda(), 64, res=xc, act=F_.ACT_NONE, alpha=0.2)), kind
    t = F_.conv3x3(xc, wp1, b1.cuda(), 64, act=F_.ACT_RELU)
    dm = (mid.float() - t.float()).abs()
    assert float(dm.max()) <= 2.0 ** -7 * max(1.0, float(t.float().abs().max())) and float((dm > 0).float().mean()) <= 0.02, kind
    assert bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("shape", [(1, 20, 70), (2, 37, 130), (1, 5, 3), (1, 9, 125), (1, 64, 62), (3, 1, 1), (1, 70, 190)])
def test_pair_sign_bits_forward_and_block_data_gradient(F_, shape):
    """SRResidualBlock under training (core/models/model.py:22-33, core/trainer.py:160).  Forward: the launch that keeps
    `mid` also writes its ReLU pattern as one bit per element -- outputs unchanged.  Backward: ONE launch on transposed /
    flipped packs, dL/d(out) and those bits gives the masked gradient of `mid` and dL/d(in); both equal, bit for bit, the
    two single-layer launches it replaces (data gradient of the second layer masked by the stored `mid`, data gradient
    of the first layer scaled by 0.2 plus the skip path)."""
    n, h, w = shape
    x, w1, b1, w2, b2 = _case(900 * h + w + n, n, h, w)
    xc = x.cuda()
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    out0, mid0 = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2, keep_mid=True)
    out, mid, bits = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2, emit_bits=True)
    assert torch.equal(out, out0) and torch.equal(mid, mid0)
    g = torch.Generator().manual_seed(h * 7 + w)
    dz = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16).cuda()
    wtb = F_.conv3x3_pack_weights(w2.cuda(), 64, 64, transpose_flip=True)
    wta = F_.conv3x3_pack_weights(w1.cuda(), 64, 64, transpose_flip=True)
    dx, g2 = F_.conv3x3_pair(dz, wtb, None, wta, None, act2=F_.ACT_NONE, alpha=0.2, apply_bits=bits)
    g2_ref = F_.conv3x3(dz, wtb, None, 64, res=mid, act=F_.ACT_NONE, res_mode=F_.RES_MASK)
    dx_ref = F_.conv3x3(g2_ref, wta, None, 64, res=dz, act=F_.ACT_NONE, alpha=0.2)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx.float()).all())
    assert torch.equal(g2, g2_ref), float((g2.float() - g2_ref.float()).abs().max())
    assert torch.equal(dx, dx_ref), float((dx.float() - dx_ref.float()).abs().max())
    # against the oracle's statement of the same two products
    from oracle import oracle as orc
    dzn = dz.float().permute(0, 3, 1, 2).cpu().numpy()
    midn = mid.float().permute(0, 3, 1, 2).cpu().numpy()
    g2o = orc.conv2d_bwd_input(dzn, w2.numpy(), midn.shape, padding=1) * (midn > 0)
    assert np.abs(g2.float().permute(0, 3, 1, 2).cpu().numpy() - g2o).max() <= 2e-2


def test_pair_sign_bits_full_size(F_):
    """The training shape (range partition, 17 strips x 16 images) and 4K (two-class partition): bits written by one
    partition's segments are read by the backward launch's, halo rows and columns included."""
    _, w1, b1, w2, b2 = _case(3, 1, 8, 8)
    wp1, wp2 = F_.conv3x3_pack_weights(w1.cuda(), 64, 64), F_.conv3x3_pack_weights(w2.cuda(), 64, 64)
    wtb = F_.conv3x3_pack_weights(w2.cuda(), 64, 64, transpose_flip=True)
    wta = F_.conv3x3_pack_weights(w1.cuda(), 64, 64, transpose_flip=True)
    for n, h, w in ((16, 1024, 1024), (1, 2160, 3840)):
        g = torch.Generator(device="cuda").manual_seed(h + n)
        xc = (torch.rand(n, h, w, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        dz = (torch.rand(n, h, w, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        out, mid, bits = F_.conv3x3_pair(xc, wp1, b1.cuda(), wp2, b2.cuda(), act2=F_.ACT_NONE, alpha=0.2, emit_bits=True)
        dx, g2 = F_.conv3x3_pair(dz, wtb, None, wta, None, act2=F_.ACT_NONE, alpha=0.2, apply_bits=bits)
        g2_ref = F_.conv3x3(dz, wtb, None, 64, res=mid, act=F_.ACT_NONE, res_mode=F_.RES_MASK)
        assert torch.equal(g2, g2_ref), (n, h, w)
        del g2
        dx_ref = F_.conv3x3(g2_ref, wta, None, 64, res=dz, act=F_.ACT_NONE, alpha=0.2)
        assert torch.equal(dx, dx_ref), (n, h, w)
        del xc, dz, out, mid, bits, dx, g2_ref, dx_ref
        torch.cuda.empty_cache()
