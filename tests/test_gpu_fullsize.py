"""Full-size (BASELINE configuration 2: 540x960 -> 2160x3840) property tests.  The CPU oracle cannot run
these sizes in test time, so they check size-independent properties that tie the full-size result to
small cases the oracle does cover: crop consistency (a convolution / the whole network evaluated on a crop
with enough halo reproduces the full-frame result in the interior), determinism and frame sharding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 2160, 3840


@pytest.fixture(scope="module")
def F_():
    from superresolution_amd import functional
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return functional


def test_conv3x3_4k_crop_consistency_bit_exact(F_):
    """64->64 3x3 at 2160x3840: any interior window equals the same layer run on that window plus a 1-pixel
    halo, bit for bit (same MFMA accumulation order per pixel whatever the strip / segment decomposition)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand(1, H, W, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g) * 0.1
    wp = F_.conv3x3_pack_weights(w, 64, 64)
    full = F_.conv3x3(x, wp, b, 64, act=F_.ACT_RELU)
    for (y0, y1, x0, x1) in [(1000, 1100, 2000, 2300), (0, 40, 0, 200), (2100, 2160, 3700, 3840), (500, 517, 100, 229)]:
        ya, yb, xa, xb = max(y0 - 1, 0), min(y1 + 1, H), max(x0 - 1, 0), min(x1 + 1, W)
        crop = F_.conv3x3(x[:, ya:yb, xa:xb].contiguous(), wp, b, 64, act=F_.ACT_RELU)
        # rows/columns that touch an artificial crop border are excluded; true image borders are kept
        iy0, ix0 = y0 - ya, x0 - xa
        assert torch.equal(crop[:, iy0:iy0 + (y1 - y0), ix0:ix0 + (x1 - x0)], full[:, y0:y1, x0:x1])


@pytest.mark.parametrize("size", [(1, H, W), (16, 1024, 1024)])
def test_conv3x3_pair_fullsize_crop_consistency_bit_exact(F_, size):
    """The fused residual pair at 2160x3840 and at the training size: any interior window equals the same pair run on
    that window plus a 2-pixel halo, bit for bit, whatever strip / segment / pipeline position computed it; two runs
    agree bit for bit; the form that also writes the intermediate map gives the same output."""
    n, h, w = size
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(n, h, w, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    w1 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    w2 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b1 = torch.randn(64, device="cuda", generator=g) * 0.1
    b2 = torch.randn(64, device="cuda", generator=g) * 0.1
    wp1, wp2 = F_.conv3x3_pack_weights(w1, 64, 64), F_.conv3x3_pack_weights(w2, 64, 64)
    full = F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2)
    assert bool(torch.isfinite(full.float()).all())
    assert torch.equal(full, F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2))
    full_m, mid = F_.conv3x3_pair(x, wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2, keep_mid=True)
    assert torch.equal(full, full_m)
    ref_mid = F_.conv3x3(x, wp1, b1, 64, act=F_.ACT_RELU)
    assert float((mid.float() - ref_mid.float()).abs().max()) <= 2.0 ** -7 * float(ref_mid.float().abs().max())
    assert torch.equal(full, F_.conv3x3(mid, wp2, b2, 64, res=x, act=F_.ACT_NONE, alpha=0.2))
    wins = [(h // 2 - 50, h // 2 + 50, w // 2 - 100, w // 2 + 200), (0, 40, 0, 200), (h - 60, h, w - 140, w), (500, 517, 100, 229)]
    for img in sorted({0, n - 1}):
        for (y0, y1, x0, x1) in wins:
            ya, yb, xa, xb = max(y0 - 2, 0), min(y1 + 2, h), max(x0 - 2, 0), min(x1 + 2, w)
            crop = F_.conv3x3_pair(x[img:img + 1, ya:yb, xa:xb].contiguous(), wp1, b1, wp2, b2, act2=F_.ACT_NONE, alpha=0.2)
            iy0, ix0 = y0 - ya, x0 - xa
            assert torch.equal(crop[:, iy0:iy0 + (y1 - y0), ix0:ix0 + (x1 - x0)], full[img:img + 1, y0:y1, x0:x1])


def test_supersampler_4k_crop_consistency_and_determinism():
    """Whole network at 540x960 -> 2160x3840: deterministic, and an LR crop with a 16-pixel LR halo
    (64 HR pixels > the receptive field of 3 LR + 25 HR layers) reproduces the full frame in its interior."""
    from superresolution_amd import SuperSampler
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 1).cuda().eval()
    g = torch.Generator().manual_seed(9)
    lr = torch.rand(1, 3, 540, 960, generator=g).cuda()
    hb = torch.rand(1, 10, H, W, generator=g).cuda()
    full = m.infer(lr, hb)
    assert full.shape == (1, 3, H, W) and bool(torch.isfinite(full).all())
    assert torch.equal(full, m.infer(lr, hb))
    # crop: LR rows 200..296, cols 400..528 (+16 halo each side) -> HR interior
    ly0, ly1, lx0, lx1, halo = 200, 296, 400, 528, 16
    lrc = lr[:, :, ly0 - halo:ly1 + halo, lx0 - halo:lx1 + halo].contiguous()
    hbc = hb[:, :, 4 * (ly0 - halo):4 * (ly1 + halo), 4 * (lx0 - halo):4 * (lx1 + halo)].contiguous()
    crop = m.infer(lrc, hbc)
    a = crop[:, :, 4 * halo:-4 * halo, 4 * halo:-4 * halo]
    b = full[:, :, 4 * ly0:4 * ly1, 4 * lx0:4 * lx1]
    # the fixed-ratio up-sampling coordinates of a crop differ from the full frame's in the last fp32 bit, and
    # one bf16 rounding of a feature can flip: bounded by the bf16 tolerance, far inside it in practice
    assert (a - b).abs().max().item() <= 2e-2
    assert (a - b).abs().mean().item() <= 1e-3


def test_frame_sharding_equals_sequential():
    """Configuration 5 (frames sharded by rank): the frames a rank owns give the same output as when one
    process runs them all, because nothing in the path couples frames."""
    from superresolution_amd import SuperSampler
    from superresolution_amd.dp import shard_range
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 1).cuda().eval()
    g = torch.Generator().manual_seed(11)
    lr = torch.rand(4, 3, 135, 240, generator=g).cuda()
    hb = torch.rand(4, 10, 540, 960, generator=g).cuda()
    batched = m.infer(lr, hb)
    world = 2
    for rank in range(world):
        for i in shard_range(4, rank, world):
            assert torch.equal(m.infer(lr[i:i + 1], hb[i:i + 1]), batched[i:i + 1])


def test_training_step_at_bench_size_stays_finite_and_learns():
    """BASELINE configuration 3 (16 crops of 1024x1024): three optimiser steps keep the output, the loss and
    every gradient finite, the loss falls, and the first step's gradients repeat bit for bit."""
    from superresolution_amd import SuperSampler, functional as F_
    from superresolution_amd.dp import FlatGradAllReducer
    from superresolution_amd.loss import FusedLossGroup
    dev = torch.device("cuda:0")
    B, S = 16, 1024
    g = torch.Generator().manual_seed(42)
    alias = torch.rand(B, 3, S, S, generator=g).to(dev)
    target = torch.rand(B, 3, S, S, generator=g).to(dev)
    buffers = torch.rand(B, 1, 10, S, S, generator=g).to(dev)
    lg = FusedLossGroup(1.0, 0.5, clip=(0.0, 10.0), report=False)

    def run(steps):
        torch.manual_seed(42)
        model = SuperSampler(3, 10, 1).to(dev).train()
        red = FlatGradAllReducer(model.parameters())
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        losses, first = [], None
        for _ in range(steps):
            lr = F_.nearest_down(alias, 0.25).view(B, 1, 3, S // 4, S // 4)
            batch = model({"scale": 4.0, "hr_frames": target, "lr_frames": lr, "hr_buffers": buffers})
            red.zero_grad()
            loss, _ = lg(batch)
            loss.backward()
            assert bool(torch.isfinite(batch["sr_frames"]).all()) and bool(torch.isfinite(red.flat).all())
            if first is None:
                first = red.flat.clone()
            losses.append(float(loss.detach()))
            opt.step()
        return losses, first

    losses, g0 = run(3)
    assert all(np.isfinite(losses)) and losses[2] < losses[1] < losses[0]
    _, g1 = run(1)
    assert torch.equal(g0, g1)


def _crop_check(F_, x, run, windows, n_check=(0,), xb=None, res=None, nchw=False):
    """`run(x, xb, res)` on the full tensors and on windows with a one-pixel halo: interiors must be bit-equal."""
    full = run(x, xb, res)
    n, hh, ww = x.shape[:3]
    for img in n_check:
        for (y0, y1, x0, x1) in windows:
            ya, yb, xa, xe = max(y0 - 1, 0), min(y1 + 1, hh), max(x0 - 1, 0), min(x1 + 1, ww)
            cut = lambda t: None if t is None else t[img:img + 1, ya:yb, xa:xe].contiguous()
            crop = run(cut(x), cut(xb), cut(res))
            iy0, ix0 = y0 - ya, x0 - xa
            if nchw:
                a, b = crop[:, :, iy0:iy0 + (y1 - y0), ix0:ix0 + (x1 - x0)], full[img:img + 1, :, y0:y1, x0:x1]
            else:
                a, b = crop[:, iy0:iy0 + (y1 - y0), ix0:ix0 + (x1 - x0)], full[img:img + 1, y0:y1, x0:x1]
            assert torch.equal(a, b), f"image {img} window {(y0, y1, x0, x1)}"


@pytest.mark.parametrize("case", ["res_add", "res_mask", "concat", "c16", "to3_f32", "c48", "c32"])
def test_conv3x3_variants_training_size_crop_consistency(F_, case):
    """Every kernel configuration the two networks use, at the training batch (16 x 1024 x 1024, several work
    items per workgroup) : windows of any image equal the same layer run on the window alone, bit for bit.
    Catches races that only appear under full memory load (they never showed in the small parity cases)."""
    g = torch.Generator(device="cuda").manual_seed(17)
    n, s = 16, 1024
    cfg = {"res_add": (64, 0, 64), "res_mask": (64, 0, 64), "concat": (32, 64, 64), "c16": (16, 0, 64),
           "to3_f32": (64, 0, 3), "c48": (48, 0, 48), "c32": (32, 0, 32)}[case]
    ca, cb, cout = cfg
    x = torch.relu(torch.rand(n, s, s, ca, device="cuda", generator=g) - 0.3).to(torch.bfloat16)
    xb = torch.relu(torch.rand(n, s, s, cb, device="cuda", generator=g) - 0.3).to(torch.bfloat16) if cb else None
    w = torch.randn(cout, ca + cb, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    wp = F_.conv3x3_pack_weights(w, ca, ca, cb, cb)
    res = None
    if case in ("res_add", "res_mask"):
        res = (torch.rand(n, s, s, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    kw = dict(act=F_.ACT_RELU)
    if case == "res_add":
        kw = dict(act=F_.ACT_NONE, alpha=0.2, res_mode=F_.RES_ADD)
    if case == "res_mask":
        kw = dict(act=F_.ACT_NONE, res_mode=F_.RES_MASK)
    if case == "to3_f32":
        kw = dict(act=F_.ACT_NONE, out_nchw_f32=True)
    run = lambda x_, xb_, r_: F_.conv3x3(x_, wp, b, cout, src_b=xb_, res=r_, **kw)
    windows = [(0, 30, 0, 140), (500, 540, 300, 600), (990, 1024, 900, 1024), (255, 258, 127, 130)]
    _crop_check(F_, x, run, windows, n_check=(0, 7, 15), xb=xb, res=res, nchw=(case == "to3_f32"))
    # and the launch repeats bit for bit
    assert torch.equal(run(x, xb, res), run(x, xb, res))


def test_wgrad_training_size_partition_and_determinism(F_):
    """Weight / bias gradient at 16 x 1024 x 1024: repeats bit for bit, and equals the sum over four batch
    quarters (accumulate form) up to fp32 summation order."""
    g = torch.Generator(device="cuda").manual_seed(23)
    n, s = 16, 1024
    x = torch.relu(torch.rand(n, s, s, 64, device="cuda", generator=g) - 0.4).to(torch.bfloat16)
    dz = ((torch.rand(n, s, s, 64, device="cuda", generator=g) - 0.5) * 1e-2).to(torch.bfloat16)
    dw, db = torch.empty(64, 64, 3, 3, device="cuda"), torch.empty(64, device="cuda")
    F_.conv3x3_wgrad(x, dz, dw, db, 64, 64)
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    F_.conv3x3_wgrad(x, dz, dw2, db2, 64, 64)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    pw, pb = torch.zeros_like(dw), torch.zeros_like(db)
    for q in range(4):
        F_.conv3x3_wgrad(x[4 * q:4 * q + 4], dz[4 * q:4 * q + 4], pw, pb, 64, 64, accumulate=True)
    assert bool(torch.isfinite(dw).all())
    assert (pw - dw).abs().max().item() <= 1e-4 * dw.abs().max().item()
    assert (pb - db).abs().max().item() <= 1e-4 * db.abs().max().item()


def test_reprfn_full_size_determinism_and_batching():
    """RepRFN (configs/trainer/4x_RepRFN.yaml) at 270x480 -> 1080x1920: finite, repeats bit for bit, and a batch
    of two equals the frames run one at a time (its 48-channel layers run two waves per SIMD)."""
    from superresolution_amd import RepRFN
    torch.manual_seed(42)
    m = RepRFN().cuda().eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 270, 480, generator=g).cuda()
    with torch.no_grad():
        y = m.infer(x)
        assert y.shape == (2, 3, 1080, 1920) and bool(torch.isfinite(y).all()) and float(y.abs().max()) < 1e3
        assert torch.equal(y, m.infer(x))
        for i in range(2):
            assert torch.equal(m.infer(x[i:i + 1]), y[i:i + 1])


def test_temporal_variant_4k_is_finite_and_reproducible():
    """Temporal SuperSampler at 544x960 -> 2176x3840 (every launch shape of the composed wide / strided / 1x1 /
    transposed layers at production size): finite, bit-reproducible, and the in-place depth compression leaves the
    caller's buffers as the reference does (depth channel rewritten, the rest untouched)."""
    from superresolution_amd.temporal import SuperSampler as Temporal
    torch.manual_seed(42)
    m = Temporal(3, 10, 4, 1).cuda().eval()
    g = torch.Generator().manual_seed(3)
    h, w = 544, 960
    base = [torch.rand(1, 3, h, w, generator=g), torch.rand(1, 10, 4 * h, 4 * w, generator=g), torch.rand(1, 10, h, w, generator=g),
            torch.randn(1, 1, 2, h, w, generator=g), torch.rand(1, 1, 3, h, w, generator=g), torch.rand(1, 1, 10, h, w, generator=g)]
    base[1][:, 9] *= 8.0    # depths above both compression thresholds
    run = lambda: m.infer(*[t.clone().cuda() for t in base])
    sr, ew = run()
    sr2, ew2 = run()
    assert sr.shape == (1, 3, 4 * h, 4 * w) and bool(torch.isfinite(sr).all()) and float(sr.abs().max()) < 1e3
    assert torch.equal(sr, sr2) and torch.equal(ew, ew2)
    hb = base[1].clone().cuda()
    m.infer(base[0].cuda(), hb, base[2].clone().cuda(), base[3].cuda(), base[4].cuda(), base[5].clone().cuda())
    assert torch.equal(hb[:, :9].cpu(), base[1][:, :9]) and float(hb[:, 9].max()) < 10.0 and not torch.equal(hb[:, 9].cpu(), base[1][:, 9])


def test_warp_4k_three_channels_zero_and_integer_motion():
    """backward_warping (core/tools/math.py:10-17) of a 3-channel 2160x3840 frame, properties that need no oracle:
    nearest mode with zero motion is the identity and with an integer shift a shifted copy with a zero border, bit for
    bit; bilinear mode lands within the rounding of the reference's normalise / un-normalise arithmetic (the sampling
    position of pixel x is x + m up to 1e-4 pixel, so the neighbour's weight is at most that)."""
    from superresolution_amd import functional as F_
    g = torch.Generator().manual_seed(3)
    h, w = 2160, 3840
    fr = torch.rand(1, 3, h, w, generator=g).cuda()
    zero = torch.zeros(1, 2, h, w, device="cuda")
    assert torch.equal(F_.backward_warp(fr, zero, "nearest", True), fr)
    assert float((F_.backward_warp(fr, zero, "bilinear", True) - fr).abs().max()) <= 5e-4
    for dx, dy in ((7, -3), (-64, 40)):
        mv = torch.zeros(1, 2, h, w, device="cuda")
        mv[:, 0], mv[:, 1] = float(dx), float(dy)
        ref = torch.zeros_like(fr)
        ys, ye = max(0, -dy), min(h, h - dy)
        xs, xe = max(0, -dx), min(w, w - dx)
        ref[:, :, ys:ye, xs:xe] = fr[:, :, ys + dy:ye + dy, xs + dx:xe + dx]
        assert torch.equal(F_.backward_warp(fr, mv, "nearest", True), ref)
        got = F_.backward_warp(fr, mv, "bilinear", True)
        inner = (slice(None), slice(None), slice(ys + 1, ye - 1), slice(xs + 1, xe - 1))
        assert float((got[inner] - ref[inner]).abs().max()) <= 5e-4
        outside = torch.ones_like(fr, dtype=torch.bool)
        outside[:, :, max(0, ys - 1):ye + 1, max(0, xs - 1):xe + 1] = False
        assert float(got[outside].abs().max()) == 0.0
    # the bf16 NHWC form gives the same picture as the fp32 form on the same (bf16-representable) values
    fr16 = fr[:, :, :540, :960].to(torch.bfloat16)
    nhwc = torch.zeros(1, 540, 960, 8, dtype=torch.bfloat16, device="cuda")
    nhwc[..., :3] = fr16.permute(0, 2, 3, 1)
    mv = (torch.randn(1, 2, 540, 960, generator=g) * 4).cuda()
    a = F_.backward_warp_nhwc(nhwc, mv, "bilinear", True)[..., :3].permute(0, 3, 1, 2).float()
    b = F_.backward_warp(fr16.float().contiguous(), mv, "bilinear", True)
    assert torch.equal(a, b.to(torch.bfloat16).float())


@pytest.mark.parametrize("ratio", [2, 4])
@pytest.mark.parametrize("shape", [(1, 5, 37, 8), (2, 33, 64, 32), (1, 1, 1, 16), (1, 270, 480, 32)])
def test_integer_ratio_upsample_equals_the_general_sampler(ratio, shape):
    """The R x R-outputs-per-thread form of the fixed-ratio up-sampling (core/models/model.py:153,162) against the
    fp32 sampler on the same bf16 values: same taps, same weights, same order of additions, so after rounding to bf16
    the maps are equal bit for bit; a non-integer ratio takes the general bf16 kernel and must agree with it too."""
    from superresolution_amd import functional as F_
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h * w + c)
    x = (torch.rand(n, h, w, c, generator=g) - 0.5).to(torch.bfloat16).cuda()
    got = F_.upsample_grid_bilinear_nhwc(x, ratio * h, ratio * w)
    ref = F_.upsample_grid_bilinear(x.float().permute(0, 3, 1, 2).contiguous(), ratio * h, ratio * w)
    assert torch.equal(got.permute(0, 3, 1, 2).float(), ref.to(torch.bfloat16).float())
    if h > 1:
        got3 = F_.upsample_grid_bilinear_nhwc(x, 3 * h, 3 * w)           # general kernel
        ref3 = F_.upsample_grid_bilinear(x.float().permute(0, 3, 1, 2).contiguous(), 3 * h, 3 * w)
        assert torch.equal(got3.permute(0, 3, 1, 2).float(), ref3.to(torch.bfloat16).float())
