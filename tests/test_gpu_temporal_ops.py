"""GPU parity of the temporal variant's operators (core/models/model_final.py): transposed 2x2/stride-2
convolution, x2 bilinear up-sampling with align_corners=True, GradientMap, and its convolution shapes
(1x1, 3x3 stride 2, wide 3x3) through the auto-dispatching convolution, against outputs of the reference's
own modules (tests/golden/temporal_ops.npz) and the CPU oracle.  bf16 outputs: <= 2e-2."""
import os

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


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "temporal_ops.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_gradient_map_golden(F_, z):
    y = F_.gradient_map(dev(z["grad.x"])).cpu().numpy()
    assert np.abs(y - z["grad.y"]).max() <= 2e-6
    rng = np.random.default_rng(1)
    x = rng.random((3, 6, 40, 57), dtype=np.float32)
    assert np.abs(F_.gradient_map(dev(x)).cpu().numpy() - orc.gradient_map(x)).max() <= 2e-6
    for shape in [(1, 1, 1, 1), (2, 3, 1, 300), (1, 2, 301, 1), (1, 1, 5, 257), (2, 2, 9, 4), (1, 3, 33, 1028)]:   # single rows / columns, rows wider than a block, both kernels
        x = rng.random(shape, dtype=np.float32)
        assert np.abs(F_.gradient_map(dev(x)).cpu().numpy() - orc.gradient_map(x)).max() <= 2e-6


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("name", ["t64_64", "t192_192"])
def test_conv_transpose_golden(F_, z, name):
    x, w, b = z[f"convT.{name}.x"], z[f"convT.{name}.w"], z[f"convT.{name}.b"]
    y = F_.conv_transpose2x2s2(F_.nchw_f32_to_nhwc_bf16(dev(x)), dev(w), dev(b), x.shape[1], act=F_.ACT_RELU)
    got = F_.nhwc_bf16_to_nchw_f32(y, w.shape[1]).cpu().numpy()
    assert got.shape == z[f"convT.{name}.y"].shape
    assert np.abs(got - z[f"convT.{name}.y"]).max() <= 2e-2
    assert np.abs(got - orc.conv_transpose2x2s2(x, w, b, relu=True)).max() <= 2e-2


@pytest.mark.parametrize("name", ["s2_32_48", "s2_192_256", "p1_160_160", "p1_112_64", "c3_160_128", "c3_24_64"])
def test_temporal_conv_shapes_golden(F_, z, name):
    key = "tconv." + name
    k, st, pd = (int(v) for v in z[key + ".geom"])
    x, w, b = z[key + ".x"], z[key + ".w"], z[key + ".b"]
    y = F_.conv2d_auto(F_.nchw_f32_to_nhwc_bf16(dev(x)), dev(w), dev(b), stride=st, padding=pd, act=F_.ACT_RELU)
    got = F_.nhwc_bf16_to_nchw_f32(y, w.shape[0]).cpu().numpy()
    assert got.shape == z[key + ".y"].shape
    assert np.abs(got - z[key + ".y"]).max() <= 2e-2
    # the fp32-accumulate bar: every one of these layers is ONE launch that accumulates its whole reduction in fp32, so
    # against the oracle on the SAME bf16-rounded operands only the final bf16 rounding of the result remains
    xr, wr = _bf16(x), _bf16(w)
    ref = orc.conv2d(xr, wr, b, st, pd, act=1)
    assert (np.abs(got - ref) <= 1e-3 + 2.0 ** -8 * np.abs(ref)).all()


def test_conv2d_auto_wide_output_uses_mfma_slices(F_):
    """3x3 with 64 inputs and 128 outputs: two MFMA launches over cout_off, against the oracle."""
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((1, 64, 20, 33)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    w = torch.from_numpy((rng.standard_normal((128, 64, 3, 3)) * 0.05).astype(np.float32)).to(torch.bfloat16).float().numpy()
    b = (rng.standard_normal(128) * 0.1).astype(np.float32)
    y = F_.conv2d_auto(F_.nchw_f32_to_nhwc_bf16(dev(x)), dev(w), dev(b), act=F_.ACT_RELU)
    assert y.shape == (1, 20, 33, 128)
    got = F_.nhwc_bf16_to_nchw_f32(y, 128).cpu().numpy()
    assert np.abs(got - orc.conv2d(x, w, b, 1, 1, act=1)).max() <= 2e-2


def test_upsample_x2_align_corners_golden(F_, z):
    x = z["up2.x"]
    y = F_.resize_bilinear_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(x)), 2 * x.shape[2], 2 * x.shape[3], align_corners=True)
    got = F_.nhwc_bf16_to_nchw_f32(y, x.shape[1]).cpu().numpy()
    assert np.abs(got - z["up2.y"]).max() <= 1e-2
    rng = np.random.default_rng(3)
    x2 = torch.from_numpy(rng.standard_normal((1, 32, 7, 11)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    y2 = F_.resize_bilinear_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(x2)), 14, 22, align_corners=True)
    assert np.abs(F_.nhwc_bf16_to_nchw_f32(y2, 32).cpu().numpy() - orc.resize_bilinear_align_corners(x2, 14, 22)).max() <= 2e-2


def test_disocclusion_kernel_matches_reference_masks_exactly():
    """Fused disocclusion kernel against the reference's own mask functions (tests/golden/temporal_masks.npz):
    comparisons and selections only, so every output is bit-exact."""
    from superresolution_amd import functional as F_
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "temporal_masks.npz"))
    for t in "abc":
        args = [torch.from_numpy(z[f"{t}.{k}"]).cuda() for k in ("lr_frame", "lr_buffer", "warped_frames", "warped_buffer")]
        marked, aux, ew, mask = F_.disocclusion(*args, want_mask=True)
        assert np.array_equal(mask.cpu().numpy(), z[t + ".mask"])
        assert np.array_equal(marked.cpu().numpy(), z[t + ".marked_frames"])
        assert np.array_equal(aux.cpu().numpy(), z[t + ".aux"])
        assert np.array_equal(ew.cpu().numpy(), z[t + ".enhance_weight"])
        d = F_.clamp_depth(torch.from_numpy(z[t + ".depth"]).cuda())
        assert np.abs(d.cpu().numpy() - z[t + ".depth_clamped"]).max() <= 1e-6


def test_disocclusion_kernel_vs_oracle_larger_shape():
    from oracle import oracle as orc
    from superresolution_amd import functional as F_
    g = torch.Generator().manual_seed(12)
    b, f, h, w = 2, 3, 67, 131
    smooth = lambda c: torch.nn.functional.interpolate(torch.rand(b, c, 5, 7, generator=g), size=(h, w), mode="bilinear")
    cf, cb = smooth(3), smooth(10)
    cb[:, 9] *= 4.0
    hf = (cf[:, None] + (torch.rand(b, f, 3, h, w, generator=g) - 0.5) * 0.3).clamp(0, 1)
    hb = cb[:, None] + (torch.rand(b, f, 10, h, w, generator=g) - 0.5) * 0.6
    om, omf, oaux, oew = orc.disocclusion(cf.numpy(), cb.numpy(), hf.numpy(), hb.numpy())
    marked, aux, ew, mask = F_.disocclusion(cf.cuda(), cb.cuda(), hf.cuda(), hb.cuda(), want_mask=True)
    assert 0.05 < om.mean() < 0.95
    assert np.array_equal(mask.cpu().numpy(), om) and np.array_equal(marked.cpu().numpy(), omf)
    assert np.array_equal(aux.cpu().numpy(), oaux) and np.array_equal(ew.cpu().numpy(), oew)


@pytest.mark.parametrize("name", ["t64_64", "t192_192"])
def test_conv_transpose_on_mfma_golden(F_, z, name):
    """The transposed convolution as a 1x1 MFMA layer to 4*cout channels plus a depth-to-space copy, against the
    reference module's output and the oracle."""
    x, w, b = z[f"convT.{name}.x"], z[f"convT.{name}.w"], z[f"convT.{name}.b"]
    wp, bp = torch.nn.Parameter(dev(w)), torch.nn.Parameter(dev(b))
    for _ in range(2):  # the second call uses the cached rearranged weight
        y = F_.conv_transpose2x2s2_auto(F_.nchw_f32_to_nhwc_bf16(dev(x)), wp, bp, act=F_.ACT_RELU)
        got = F_.nhwc_bf16_to_nchw_f32(y, w.shape[1]).cpu().numpy()
        assert got.shape == z[f"convT.{name}.y"].shape
        assert np.abs(got - z[f"convT.{name}.y"]).max() <= 2e-2
        assert np.abs(got - orc.conv_transpose2x2s2(x, w, b, relu=True)).max() <= 2e-2
        ref = orc.conv_transpose2x2s2(_bf16(x), _bf16(w), b, relu=True)      # fp32-accumulate bar on bf16-rounded operands
        assert (np.abs(got - ref) <= 1e-3 + 2.0 ** -8 * np.abs(ref)).all()


def test_depth_to_space_is_a_permutation(F_):
    from superresolution_amd._lib import check, lib
    n, h, w, c = 2, 5, 7, 16
    src = torch.arange(n * h * w * 4 * c, dtype=torch.float32).remainder(251.0).view(n, h, w, 4 * c).to(torch.bfloat16).cuda()
    dst = torch.empty(n, 2 * h, 2 * w, c, dtype=torch.bfloat16, device="cuda")
    check(lib.srhip_depth_to_space2_nhwc(src.data_ptr(), dst.data_ptr(), n, h, w, c, torch.cuda.current_stream().cuda_stream))
    ref = src.view(n, h, w, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, c)
    assert torch.equal(dst, ref)


@pytest.mark.parametrize("shape", [(1, 5, 7, 64), (2, 9, 4, 32)])
def test_attention_fuse_matches_the_reference_formula(F_, shape):
    """model_final.py:422 on bf16 maps with fp32 arithmetic: the fused kernel against the torch expression."""
    n, h, w, c = shape
    g = torch.Generator().manual_seed(c)
    cur, fmap, grad = (torch.randn(n, h, w, c, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    gates = [(torch.randn(n, h, w, 16, generator=g) * 2).to(torch.bfloat16).cuda()[..., 0:1] for _ in range(4)]
    f_self, g_cross, g_self, f_cross = gates
    ref = cur.float() + torch.sigmoid(f_self.float() * g_cross.float()) * fmap.float() \
        + torch.sigmoid(g_self.float() * f_cross.float()) * grad.float()
    got = F_.attention_fuse(cur, fmap, grad, f_self, g_cross, g_self, f_cross)
    assert got.shape == cur.shape and got.dtype == torch.bfloat16
    # one bf16 rounding of an fp32 result whose sigmoid may differ from torch's by an ulp
    assert (got.float() - ref).abs().max().item() <= 2e-2
    assert (got.float() - ref.to(torch.bfloat16).float()).abs().gt(0).float().mean().item() <= 0.01
    with pytest.raises(ValueError):
        F_.attention_fuse(cur, fmap, grad, f_self.contiguous(), g_cross, g_self, f_cross)


def test_dilate_velocity_bit_exact(F_):
    """srhip_dilate_velocity_f32 against the reference function's own outputs and, at other shapes (batch, a row wider
    than a block, 7x7 window), against the oracle."""
    from superresolution_amd.tools_math import dilate_velocity
    zz = np.load(os.path.join(os.path.dirname(__file__), "golden", "dilate_velocity.npz"))
    for tag in ("a", "b", "c"):
        got = dilate_velocity(dev(zz[f"{tag}.motion"]), dev(zz[f"{tag}.depth"]), int(zz[f"{tag}.k"])).cpu().numpy()
        assert np.array_equal(got, zz[f"{tag}.out"]), tag
    rng = np.random.default_rng(8)
    for (b, h, w, k) in [(2, 11, 300, 3), (3, 5, 4, 7), (1, 1, 1, 3)]:
        m = rng.standard_normal((b, h, w, 2)).astype(np.float32)
        d = (np.round(rng.random((b, 1, h, w)) * 6) / 2 + 0.5).astype(np.float32)
        assert np.array_equal(F_.dilate_velocity(dev(m), dev(d), k).cpu().numpy(), orc.dilate_velocity(m, d, k))
    with pytest.raises(ValueError):
        F_.dilate_velocity(dev(m), dev(d[:, :, :0]), 3)


@pytest.mark.parametrize("shape", [(1, 10, 16, 24), (2, 6, 20, 52), (3, 1, 4, 4), (1, 23, 8, 12)])
def test_fused_pixel_unshuffle_layout_bit_exact(F_, shape):
    """PixelUnshuffle(4) + NHWC bf16 conversion in one kernel == the two kernels in sequence (partial 64-pixel tiles,
    tiles that straddle rows and images); other factors and wider inputs take the pair."""
    rng = np.random.default_rng(sum(shape))
    x = dev(rng.standard_normal(shape).astype(np.float32))
    got = F_.pixel_unshuffle_to_nhwc_bf16(x, 4)
    ref = F_.nchw_f32_to_nhwc_bf16(F_.pixel_unshuffle(x, 4))
    assert got.shape == ref.shape and got.dtype == torch.bfloat16 and torch.equal(got, ref)
    x2 = dev(rng.standard_normal((1, 3, 6, 10)).astype(np.float32))
    assert torch.equal(F_.pixel_unshuffle_to_nhwc_bf16(x2, 2), F_.nchw_f32_to_nhwc_bf16(F_.pixel_unshuffle(x2, 2)))


@pytest.mark.parametrize("shape", [(1, 10, 16, 24), (2, 6, 20, 52), (1, 10, 544, 960), (1, 3, 6, 10)])
def test_fused_pixel_unshuffle_also_yields_the_maximum(F_, shape):
    """The maximum the temporal network asserts on (core/models/model_final.py:463) out of the folding pass: equal to
    torch.max bit for bit (negative-only tensors, a maximum in the last partial tile, more tiles than work-groups), NaN
    when the tensor holds one, and the folded map is the one the plain entry point writes."""
    rng = np.random.default_rng(sum(shape))
    r = 4 if shape[2] % 4 == 0 and shape[3] % 4 == 0 else 2
    for variant in ("normal", "negative", "last", "nan"):
        a = rng.standard_normal(shape).astype(np.float32)
        if variant == "negative":
            a = -np.abs(a) - 0.25
        if variant == "last":
            a[-1, -1, -1, -1] = 77.5
        if variant == "nan":
            a[0, 0, shape[2] // 2, 1] = np.nan
        x = dev(a)
        got, vmax = F_.pixel_unshuffle_to_nhwc_bf16(x, r, want_max=True)
        assert vmax.dim() == 0 and vmax.dtype == torch.float32
        if variant == "nan":
            assert bool(torch.isnan(vmax)) and not bool(vmax < 10)      # the reference's assert fires on NaN as well
            continue
        assert float(vmax) == float(x.max())
        assert torch.equal(got, F_.pixel_unshuffle_to_nhwc_bf16(x, r))


def test_disocclusion_with_spare_auxiliary_planes_and_in_place_depth_clamp(F_):
    """aux_planes=16 leaves two planes behind the fourteen for |motion| (the history refiner's input,
    model_final.py:300) without a concatenation; clamp_depth may write over its input."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "temporal_masks.npz"))
    done = 0
    for t in "abc":
        args = [torch.from_numpy(z[f"{t}.{k}"]).cuda() for k in ("lr_frame", "lr_buffer", "warped_frames", "warped_buffer")]
        if args[2].shape[0] * args[2].shape[1] != 1:
            with pytest.raises(ValueError):
                F_.disocclusion(*args, aux_planes=16)
            continue
        marked, aux, ew = F_.disocclusion(*args, aux_planes=16)
        assert tuple(aux.shape[:2]) == (1, 16) and np.array_equal(aux[:, :14].cpu().numpy(), z[t + ".aux"])
        assert np.array_equal(marked.cpu().numpy(), z[t + ".marked_frames"]) and np.array_equal(ew.cpu().numpy(), z[t + ".enhance_weight"])
        done += 1
    g = torch.Generator().manual_seed(3)
    cf, cb = torch.rand(1, 3, 9, 13, generator=g).cuda(), torch.rand(1, 10, 9, 13, generator=g).cuda()
    hf, hb = torch.rand(1, 1, 3, 9, 13, generator=g).cuda(), torch.rand(1, 1, 10, 9, 13, generator=g).cuda()
    _, aux14, _ = F_.disocclusion(cf, cb, hf, hb)
    _, aux16, _ = F_.disocclusion(cf, cb, hf, hb, aux_planes=16)
    assert torch.equal(aux16[:, :14], aux14)
    d = (torch.rand(2, 33, 17, generator=g) * 12).cuda()
    want = F_.clamp_depth(d)
    assert F_.clamp_depth(d, out=d) is d and torch.equal(d, want)
    with pytest.raises(ValueError):
        F_.clamp_depth(want, out=want[:, ::2])


@pytest.mark.parametrize("shape", [(1, 10, 16, 24), (2, 10, 20, 52), (1, 10, 544, 960), (3, 4, 4, 4)])
def test_fused_gradient_map_unshuffle_bit_exact(F_, shape):
    """GradientMap + PixelUnshuffle(4) + NHWC bf16 conversion in one pass == the three kernels in sequence, bit for bit
    (image borders, 64-pixel tiles that straddle rows and images, planes taken out of order)."""
    rng = np.random.default_rng(sum(shape))
    x = dev(rng.standard_normal(shape).astype(np.float32))
    for planes in ((0, 1, 2, 6, 7, 8), (3,), (2, 0)):
        planes = tuple(k % shape[1] for k in planes)
        got = F_.gradient_map_unshuffle_nhwc(x, planes, 1e-6)
        ref = F_.pixel_unshuffle_to_nhwc_bf16(F_.gradient_map(x[:, list(planes)].contiguous(), 1e-6), 4)
        assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))
    with pytest.raises(ValueError):
        F_.gradient_map_unshuffle_nhwc(x, (shape[1],))


@pytest.mark.parametrize("shape,c", [((1, 5, 7, 48), 3), ((2, 9, 4, 64), 4), ((1, 136, 240, 48), 3), ((1, 3, 3, 16), 1)])
def test_pixel_shuffle_into_the_padded_nhwc_layout_is_a_permutation(F_, shape, c):
    """PixelShuffle(4) between two convolutions without the fp32 NCHW detour: equal bit for bit to the fused NHWC -> NCHW
    fp32 head followed by the layout conversion (both exact on bf16 values), padding channels zero."""
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(*shape, generator=g).to(torch.bfloat16).cuda()
    got = F_.pixel_shuffle4_nhwc_to_nhwc16(x, c)
    ref = F_.nchw_f32_to_nhwc_bf16(F_.pixel_shuffle_nhwc_to_nchw_f32(x, c, 4), 16)
    assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))
    assert not got[..., c:].any()
    with pytest.raises(ValueError):
        F_.pixel_shuffle4_nhwc_to_nhwc16(x, 5)


@pytest.mark.parametrize("shape", [(1, 5, 7, 48), (2, 19, 45, 48), (1, 136, 240, 64), (1, 8, 32, 48), (1, 1, 1, 48)])
def test_fused_tail_pixel_shuffle_conv_relu(F_, shape):
    """PixelShuffle(4) -> Conv2d(3,3,3,1,1) -> ReLU (core/models/model_final.py:446-449) in one pass against torch's own
    fp32 operators on the same bf16 map (tile borders, image borders, partial tiles): fp32 accumulation bar."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(shape[1] * 7 + shape[2])
    x = torch.randn(*shape, generator=g).to(torch.bfloat16)
    wgt = torch.randn(3, 3, 3, 3, generator=g) * 0.3
    bias = torch.randn(3, generator=g) * 0.1
    for act, fn in ((F_.ACT_RELU, torch.relu), (F_.ACT_NONE, lambda v: v)):
        got = F_.pixel_shuffle4_conv3x3_c3(x.cuda(), wgt.cuda(), bias.cuda(), act=act)
        ref = fn(TF.conv2d(TF.pixel_shuffle(x[..., :48].float().permute(0, 3, 1, 2), 4), wgt, bias, padding=1))
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert float((got.cpu() - ref).abs().max()) <= 1e-4
    got = F_.pixel_shuffle4_conv3x3_c3(x.cuda(), wgt.cuda(), None)
    assert float((got.cpu() - TF.conv2d(TF.pixel_shuffle(x[..., :48].float().permute(0, 3, 1, 2), 4), wgt, None, padding=1)).abs().max()) <= 1e-4
    with pytest.raises(ValueError):
        F_.pixel_shuffle4_conv3x3_c3(x.cuda(), wgt[:2].cuda(), bias.cuda())


@pytest.mark.parametrize("case", [((160, 160, 160, 64), False, (1, 37, 53)), ((112, 64, 64, 64), True, (2, 19, 70)),
                                  ((160, 160, 160, 64), False, (1, 1, 5)), ((112, 64, 64, 64), True, (1, 96, 130)),
                                  ((160, 160, 160, 64), False, (1, 136, 240))])
def test_conv1x1_chain_equals_the_three_layers(case):
    """The 1x1 tails of the G-buffer encoder (core/models/model_final.py:237-246, :256-258) as one launch: bit-equal to the
    three implicit-GEMM launches it replaces (same bf16 rounding of the maps between the layers, same fp32 sums in the same
    order), and within 2e-2 of the oracle's convolutions with those maps rounded to bf16."""
    from superresolution_amd import functional as F_
    chans, has_res, (n, h, w) = case
    g = torch.Generator().manual_seed(h * 31 + w)
    x = (torch.rand(n, h, w, chans[0], generator=g) - 0.3).to(torch.bfloat16).cuda()
    layers = []
    for l in range(3):
        wt = (torch.randn(chans[l + 1], chans[l], 1, 1, generator=g) * (1.5 / chans[l] ** 0.5)).cuda()
        b = (torch.randn(chans[l + 1], generator=g) * 0.1).cuda()
        layers.append((wt, b, F_.ACT_RELU))
    res = (torch.rand(n, h, w, chans[3], generator=g) - 0.5).to(torch.bfloat16).cuda() if has_res else None
    alpha = 0.45 if has_res else 1.0
    assert F_.conv1x1_chain_supported(list(chans), has_res)
    got = F_.conv1x1_chain(x, layers, alpha=alpha, res=res)
    t = x
    for l, (wt, b, act) in enumerate(layers):
        last = l == 2
        t = F_.conv2d_auto(t, wt, b, stride=1, padding=0, act=act, alpha=alpha if last else 1.0, res=res if last else None)
    torch.cuda.synchronize()
    assert got.shape == t.shape
    assert torch.equal(got, t), float((got.float() - t.float()).abs().max())
    if h * w <= 4000:
        r = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
        o = x.float().permute(0, 3, 1, 2).cpu().numpy()
        for l, (wt, b, act) in enumerate(layers):
            wq = wt.to(torch.bfloat16).float().cpu().numpy()
            o = orc.conv2d(o, wq, b.cpu().numpy(), 1, 0, act=1)
            if l < 2:
                o = r(o)
        o = alpha * o + (res.float().permute(0, 3, 1, 2).cpu().numpy() if has_res else 0.0)
        assert np.abs(got.float().permute(0, 3, 1, 2).cpu().numpy() - o).max() <= 2e-2 * max(1.0, float(np.abs(o).max()))
    # an unsupported chain is refused, not mis-run
    assert not F_.conv1x1_chain_supported([64, 64, 64, 64], False)


@pytest.mark.parametrize("case", [(2, 9, 21, 64, 16), (1, 5, 7, 32, 16), (1, 3, 50, 16, 8), (1, 4, 4, 8, 16)])
def test_attention_fuse_backward_vs_torch_autograd(case):
    """Backward of the gated sum (core/models/model_final.py:422 under autograd): against torch's autograd of the same
    formula in fp32 on the same bf16 operands; results are bf16, so the bar is one bf16 rounding (2^-8 relative) plus the
    fp32 summation order of the channel sums."""
    from superresolution_amd import autograd_ops as A_
    n, h, w, c, cg = case
    g = torch.Generator().manual_seed(c + h)
    mk = lambda *s: ((torch.rand(*s, generator=g) - 0.5) * 2).to(torch.bfloat16).cuda()
    cur, fmap, grad = mk(n, h, w, c), mk(n, h, w, c), mk(n, h, w, c)
    gates = [mk(n, h, w, cg) for _ in range(4)]
    d = mk(n, h, w, c)
    ins = [t.clone().requires_grad_(True) for t in (cur, fmap, grad, *gates)]
    out = A_.attention_fuse(*ins)
    out.backward(d)
    ref_in = [t.float().clone().requires_grad_(True) for t in (cur, fmap, grad, *gates)]
    rc, rf, rg, fs, gc, gs, fc = ref_in
    ref = rc + torch.sigmoid(fs[..., 0:1] * gc[..., 0:1]) * rf + torch.sigmoid(gs[..., 0:1] * fc[..., 0:1]) * rg
    ref.backward(d.float())
    torch.cuda.synchronize()
    assert torch.equal(out, ref.to(torch.bfloat16)) or float((out.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    for got, want in zip(ins, ref_in):
        a, b = got.grad.float(), want.grad
        assert a.shape == b.shape
        tol = 2.0 ** -8 * b.abs() + 1e-5 * float(b.abs().max()) + 1e-30
        assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() - tol).max())
    for got in ins[3:]:
        assert float(got.grad[..., 1:].abs().max()) == 0.0   # only channel 0 of a gate map carries gradient


@pytest.mark.parametrize("case", [(2, 3, 64, 96, 4), (1, 3, 30, 52, 2), (3, 1, 8, 8, 1), (1, 3, 256, 1024, 4)])
def test_enhance_weighted_l1_vs_torch(case):
    """loss = mean((1 + nearest-upsampled weight) * |sr - target|) (the temporal variant's pixel-wise loss weight,
    core/models/model_final.py:499-505) and its gradient, against the same formula in torch operators under autograd:
    value to 1e-6 relative (fp32 sums in a different order), gradient to one fp32 rounding; zero gradient where sr equals
    the target; repeatable bit for bit."""
    from superresolution_amd.loss import enhance_weighted_l1
    n, c, h, w, r = case
    g = torch.Generator().manual_seed(h + w)
    sr = torch.rand(n, c, h, w, generator=g).cuda()
    target = torch.rand(n, c, h, w, generator=g).cuda()
    target[:, :, ::3, ::5] = sr[:, :, ::3, ::5]
    ew = torch.rand(n, 1, h // r, w // r, generator=g).cuda()
    a = sr.clone().requires_grad_(True)
    loss = enhance_weighted_l1(a, target, ew)
    (loss * 3.0).backward()
    b = sr.clone().requires_grad_(True)
    wmap = torch.nn.functional.interpolate(ew, size=(h, w), mode="nearest")
    ref = ((1.0 + wmap) * (b - target).abs()).mean()
    (ref * 3.0).backward()
    a2 = sr.clone().requires_grad_(True)
    loss2 = enhance_weighted_l1(a2, target, ew)
    (loss2 * 3.0).backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
    assert float((a.grad - b.grad).abs().max()) <= 2e-7 * float(b.grad.abs().max())
    assert float(a.grad[:, :, ::3, ::5].abs().max()) == 0.0
    assert torch.equal(loss, loss2) and torch.equal(a.grad, a2.grad)


@pytest.mark.parametrize("case", [(2, 5, 7, 48, 3), (1, 3, 9, 64, 4), (1, 8, 8, 16, 1)])
def test_pixel_shuffle4_to_nhwc16_and_output_layout_are_differentiable_permutations(case):
    """nn.PixelShuffle(4) written into the next layer's 16-channel NHWC input (core/models/model_final.py:446-448) and the
    network output's NHWC bf16 -> NCHW fp32 change, each with its one-pass backward: values and gradients bit-equal to the
    torch formulation (pixel_shuffle, permute, pad, casts) on the same bf16 data."""
    from superresolution_amd import autograd_ops as A_
    n, h, w, cs, c = case
    g = torch.Generator().manual_seed(cs + c)
    t = ((torch.rand(n, h, w, cs, generator=g) - 0.5) * 4).to(torch.bfloat16).cuda()
    dy = ((torch.rand(n, 4 * h, 4 * w, 16, generator=g) - 0.5) * 4).to(torch.bfloat16).cuda()
    a = t.clone().requires_grad_(True)
    y = A_.pixel_shuffle4_to_nhwc16(a, c)
    y.backward(dy)
    b = t.clone().requires_grad_(True)
    hr = torch.nn.functional.pixel_shuffle(b[..., :16 * c].permute(0, 3, 1, 2).float(), 4)          # (n, c, 4h, 4w)
    ref = torch.nn.functional.pad(hr.permute(0, 2, 3, 1), (0, 16 - c)).to(torch.bfloat16)
    ref.backward(dy)
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and torch.equal(a.grad, b.grad)
    # output side
    x = ((torch.rand(n, 4 * h, 4 * w, 16, generator=g) - 0.5) * 4).to(torch.bfloat16).cuda()
    d = torch.rand(n, c, 4 * h, 4 * w, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    o = A_.nhwc_to_nchw_f32(xa, c)
    o.backward(d)
    xb = x.clone().requires_grad_(True)
    oref = xb[..., :c].permute(0, 3, 1, 2).float()
    oref.backward(d)
    torch.cuda.synchronize()
    assert torch.equal(o, oref) and torch.equal(xa.grad, xb.grad)


@pytest.mark.parametrize("case", [(2, 7, 9, 32), (1, 8, 8, 64), (1, 1, 5, 8)])
def test_zero_insert2_places_the_gradient_at_the_even_positions(F_, case):
    n, h, w, c = case
    ho, wo = (h + 1) // 2, (w + 1) // 2
    x = torch.randn(n, ho, wo, c).to(torch.bfloat16).cuda()
    got = F_.zero_insert2_nhwc(x, h, w)
    ref = torch.zeros(n, h, w, c, dtype=torch.bfloat16, device="cuda")
    ref[:, ::2, ::2] = x
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_scale_add_and_its_backward_vs_torch(F_):
    """bf16(a + scale * y) with a trained scalar (fmap + enhance_param * enhance, core/models/model_final.py:256-258) as one
    pass, and its one-pass backward, against the torch expression it replaces in the training graph: the forward within
    one bf16 step (fma against multiply-then-add), da bit for bit, dy within a bf16 step, dscale to 1e-4 relative; an
    element count that does not fill the last work-group."""
    from superresolution_amd import autograd_ops as A
    torch.manual_seed(5)
    for shape in ((2, 37, 41, 64), (1, 8, 8, 16)):
        a = torch.randn(shape, device="cuda").to(torch.bfloat16).requires_grad_(True)
        y = torch.randn(shape, device="cuda").to(torch.bfloat16).requires_grad_(True)
        p = torch.tensor(0.45, device="cuda", requires_grad=True)
        d = torch.randn(shape, device="cuda").to(torch.bfloat16)
        out = A.scale_add(a, y, p)
        out.backward(d)
        a2, y2, p2 = (t.detach().clone().requires_grad_(True) for t in (a, y, p))
        ref = (a2.float() + p2 * y2.float()).to(torch.bfloat16)
        ref.backward(d)
        step = lambda t: t.float().abs().clamp_min(1e-3) * 2.0 ** -7
        assert bool(((out.float() - ref.float()).abs() <= step(ref)).all())
        assert torch.equal(a.grad, a2.grad)
        assert bool(((y.grad.float() - y2.grad.float()).abs() <= step(y2.grad)).all())
        assert abs(float(p.grad) - float(p2.grad)) <= 1e-4 * max(1.0, abs(float(p2.grad)))
