"""GPU parity of the sign-bit forms of srhip_conv3x3_fwd: the ReLU patterns the backward pass of
Residual(ConvActivation, ConvActivation) / ConvActivation (core/models/model.py:17-19,61-79 under core/trainer.py:160)
needs travel as 1 bit per element, written and consumed by the convolution launches themselves.

Every form is an exact re-arrangement of launches that already exist (a convolution followed by an element-wise mask
or add on its bf16 output), so the bar is bit equality with that composition."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1, 20, 70), (2, 9, 130), (1, 33, 128), (1, 5, 3), (1, 1, 31), (1, 7, 257), (3, 2, 33)]


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def _case(seed, n, h, w, cin=64):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5).to(torch.bfloat16).cuda()
    wt = (torch.randn(64, cin, 3, 3, generator=g) * 0.06).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda()
    r = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16).cuda()
    return x, wt, b, r


def test_signbit_layout_matches_its_description(F_):
    """signbits_unpack (the documented layout) inverts what the kernel writes, on a map whose pattern is known."""
    n, h, w = 2, 3, 77
    x, wt, b, _ = _case(1, n, h, w)
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    y, bits = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)
    assert bits.dtype == torch.int32 and tuple(bits.shape) == (n, h, 3, 64)
    assert torch.equal(F_.signbits_unpack(bits, w), y > 0)
    assert 0.2 < float((y > 0).float().mean()) < 0.8


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cin", [64, 10])
def test_emit_without_residual(F_, shape, cin):
    n, h, w = shape
    x, wt, b, _ = _case(7 * h + w + cin, n, h, w, cin=F_.pad16(cin))
    wt = wt[:, :cin].contiguous()
    wp = F_.conv3x3_pack_weights(wt, cin, F_.pad16(cin))
    ref = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU)
    y, bits = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)
    assert torch.equal(y, ref)
    assert torch.equal(F_.signbits_unpack(bits, w), ref > 0)


@pytest.mark.parametrize("shape", SHAPES)
def test_emit_with_residual(F_, shape):
    """dst = relu(conv) + res in one epilogue; the bits are those of relu(conv) alone."""
    n, h, w = shape
    x, wt, b, r = _case(11 * h + w, n, h, w)
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    branch = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU)
    ref = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, res=r)
    y, bits = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, res=r, aux_mode=F_.AUX_EMIT)
    assert torch.equal(y, ref)
    assert torch.equal(F_.signbits_unpack(bits, w), branch > 0)


@pytest.mark.parametrize("shape", SHAPES)
def test_maskbits_equals_add_then_relu_mask(F_, shape):
    n, h, w = shape
    x, wt, b, r = _case(13 * h + w, n, h, w)
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    act_map, bits = F_.conv3x3(r, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)     # some ReLU output and its bits
    wt_t = F_.conv3x3_pack_weights(wt, 64, 64, cout=64, cout_off=0, transpose_flip=True)
    ref = F_.relu_mask(F_.conv3x3(x, wt_t, None, 64, res=r, alpha=0.7), act_map)
    got = F_.conv3x3(x, wt_t, None, 64, res=r, alpha=0.7, signbits=bits, aux_mode=F_.AUX_MASKBITS)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", SHAPES)
def test_dualbits_equals_conv_and_relu_mask(F_, shape):
    n, h, w = shape
    x, wt, b, r = _case(17 * h + w, n, h, w)
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    act_map, bits = F_.conv3x3(r, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)
    wt_t = F_.conv3x3_pack_weights(wt, 64, 64, cout=64, cout_off=0, transpose_flip=True)
    ref = F_.conv3x3(x, wt_t, None, 64)
    ref_masked = F_.relu_mask(ref, act_map)
    got, got_masked = F_.conv3x3(x, wt_t, None, 64, signbits=bits, aux_mode=F_.AUX_DUALBITS)
    assert torch.equal(got, ref)
    assert torch.equal(got_masked, ref_masked)


def test_signbit_forms_reject_unsupported_layers(F_):
    from superresolution_amd._lib import SrhipError
    x = torch.zeros(1, 8, 40, 32, dtype=torch.bfloat16, device="cuda")
    wp = F_.conv3x3_pack_weights(torch.zeros(64, 32, 3, 3, device="cuda"), 32, 32)
    with pytest.raises(SrhipError):
        F_.conv3x3(x, wp, None, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)            # 32 input channels: not built
    x64 = torch.zeros(1, 8, 40, 64, dtype=torch.bfloat16, device="cuda")
    wp64 = F_.conv3x3_pack_weights(torch.zeros(64, 64, 3, 3, device="cuda"), 64, 64)
    with pytest.raises(ValueError):
        F_.conv3x3(x64, wp64, None, 64, aux_mode=F_.AUX_MASKBITS, res=x64)                # no sign-bit tensor given
    with pytest.raises(ValueError):
        F_.conv3x3(x64, wp64, None, 64, aux_mode=F_.AUX_DUALBITS, signbits=torch.zeros(1, 8, 1, 64, dtype=torch.int32, device="cuda"))
    bits = F_.signbits_empty(1, 8, 40, x64.device)
    with pytest.raises(SrhipError):
        F_.conv3x3(x64, wp64, None, 64, aux_mode=F_.AUX_MASKBITS, signbits=bits)          # MASKBITS is built with a residual only
    with pytest.raises(SrhipError):
        F_.conv3x3(x64, wp64, None, 64, res=x64, res_mode=F_.RES_MASK, aux_mode=F_.AUX_EMIT)


def test_full_size_signbit_roundtrip(F_):
    """16 crops of 1024x1024 (the training configuration): the emitted bits reproduce the map's pattern, and the
    masked gradient equals the separate pass, everywhere."""
    n, h, w = 4, 1024, 1024
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16).cuda()
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda()
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    y, bits = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)
    assert torch.equal(F_.signbits_unpack(bits[:1], w), y[:1] > 0)
    assert torch.equal(F_.signbits_unpack(bits[-1:], w), y[-1:] > 0)
    d, dm = F_.conv3x3(x, wp, None, 64, signbits=bits, aux_mode=F_.AUX_DUALBITS)
    assert torch.equal(dm, F_.relu_mask(d, y))


@pytest.mark.parametrize("shape", SHAPES)
def test_signbits_from_map_equals_the_emitted_bits(F_, shape):
    n, h, w = shape
    x, wt, b, _ = _case(19 * h + w, n, h, w)
    wp = F_.conv3x3_pack_weights(wt, 64, 64)
    y, bits = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU, aux_mode=F_.AUX_EMIT)
    assert torch.equal(F_.signbits_unpack(F_.signbits_from_map(y), w), F_.signbits_unpack(bits, w))   # (bits of pixels >= w are unspecified)
    z = (torch.randn(n, h, w, 64, generator=torch.Generator().manual_seed(w)) * 0.5).to(torch.bfloat16).cuda()   # signed map
    z[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-38, -1e-38], dtype=torch.bfloat16)
    assert torch.equal(F_.signbits_unpack(F_.signbits_from_map(z), w), z > 0)
