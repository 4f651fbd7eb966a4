"""Pins the CPU oracle (oracle/) against goldens produced by the reference itself
(tests/golden/make_golden.py).  Index / permutation ops must be bit-exact; convolution and
bilinear sampling agree to 1e-5 (summation order differs from ATen's)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc


def _close(a, b, tol=1e-5):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= tol * scale, f"max abs err {err} (scale {scale})"


def test_conv_forward(ops_golden):
    z = ops_golden
    for name in z["conv.names"]:
        name = str(name)
        s, ph, pw = [int(v) for v in z[f"conv.{name}.cfg"]]
        y = orc.conv2d(z[f"conv.{name}.x"], z[f"conv.{name}.w"], z[f"conv.{name}.b"], s, (ph, pw))
        assert y.shape == z[f"conv.{name}.y"].shape, name
        _close(y, z[f"conv.{name}.y"])


@pytest.mark.parametrize("name", ["b64_64", "b10_64", "b64_3"])
def test_conv_backward(ops_golden, name):
    z = ops_golden
    x, w, dy = z[f"convbwd.{name}.x"], z[f"convbwd.{name}.w"], z[f"convbwd.{name}.dy"]
    dx = orc.conv2d_bwd_input(dy, w, x.shape, 1, 1)
    dw, db = orc.conv2d_bwd_weight(x, dy, w.shape, 1, 1)
    _close(dx, z[f"convbwd.{name}.dx"])
    _close(dw, z[f"convbwd.{name}.dw"], 2e-5)
    _close(db, z[f"convbwd.{name}.db"], 2e-5)


@pytest.mark.parametrize("hw", [(3, 4), (12, 20), (48, 80), (7, 5), (270, 480), (1080, 1920), (2160, 3840)])
def test_make_coord_bit_exact(ops_golden, hw):
    h, w = hw
    c = orc.make_coord(h, w)
    assert np.array_equal(c[0, 0, 0, :], ops_golden[f"coord.{h}x{w}.x"])
    assert np.array_equal(c[0, 1, :, 0], ops_golden[f"coord.{h}x{w}.y"])
    assert np.array_equal(c[0, 0], np.broadcast_to(c[0, 0, :1, :], (h, w)))
    assert np.array_equal(c[0, 1], np.broadcast_to(c[0, 1, :, :1], (h, w)))


@pytest.mark.parametrize("name", ["s4", "s2", "s3"])
def test_upsample_grid_bilinear(ops_golden, name):
    x, y = ops_golden[f"up.{name}.x"], ops_golden[f"up.{name}.y"]
    _close(orc.upsample_grid_bilinear(x, y.shape[2], y.shape[3]), y, 2e-6)


def test_upsample_backward(ops_golden):
    z = ops_golden
    x, dy = z["upbwd.x"], z["upbwd.dy"]
    th, tw = dy.shape[-2:]
    grid = np.ascontiguousarray(np.transpose(orc.make_coord(th, tw), (0, 2, 3, 1)))
    dx, _ = orc.grid_sample_bwd(x, grid, dy, False, need_dgrid=False)
    _close(dx, z["upbwd.dx"])


@pytest.mark.parametrize("cfg", [(3, 4, 16), (6, 5, 2), (5, 7, 3), (5, 7, 4), (270, 480, 4)])
def test_rel_coord_bit_exact(ops_golden, cfg):
    sh, sw, s = cfg
    r = orc.rel_coord(sh * s, sw * s, sh, sw)
    key = f"rel.{sh}x{sw}x{s}"
    if key in ops_golden:
        assert np.array_equal(r, ops_golden[key])
    else:
        assert np.array_equal(r[0, 0, 0, :], ops_golden[key + ".x"])
        assert np.array_equal(r[0, 1, :, 0], ops_golden[key + ".y"])


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("align", [True, False])
def test_backward_warp(ops_golden, mode, align):
    z = ops_golden
    y = orc.backward_warp(z["warp.frame"], z["warp.motion"], mode, align)
    ref = z[f"warp.{mode}.{int(align)}"]
    if mode == "nearest":
        assert np.array_equal(y, ref)
    else:
        _close(y, ref, 2e-6)


def test_warp_mv_accumulation(ops_golden):
    z = ops_golden
    mv0, mv1 = z["warp.motion"][:1], z["warp.motion"][1:]
    acc = orc.backward_warp(mv1, mv0, "bilinear", True) + mv0
    _close(acc, z["warp.accum"], 2e-6)


def test_warp_backward(ops_golden):
    z = ops_golden
    frame, motion, dy = z["warp.frame"], z["warp.motion"], z["warpbwd.dy"]
    grid = orc.warp_grid(motion)
    dframe, dgrid = orc.grid_sample_bwd(frame, grid, dy, True)
    _close(dframe, z["warpbwd.dframe"], 1e-5)
    h, w = frame.shape[-2:]
    # d grid / d motion = 2/(size-1)   (core/tools/math.py:14-15)
    dm = np.stack([dgrid[..., 0] * np.float32(2.0 / (w - 1)), dgrid[..., 1] * np.float32(2.0 / (h - 1))], axis=1)
    _close(dm, z["warpbwd.dmotion"], 2e-5)


@pytest.mark.parametrize("r", [2, 4])
def test_pixel_shuffle_bit_exact(ops_golden, r):
    z = ops_golden
    assert np.array_equal(orc.pixel_shuffle(z[f"ps.{r}.x"], r), z[f"ps.{r}.y"])
    assert np.array_equal(orc.pixel_unshuffle(z[f"pu.{r}.x"], r), z[f"pu.{r}.y"])
    assert np.array_equal(orc.pixel_unshuffle(orc.pixel_shuffle(z[f"ps.{r}.x"], r), r), z[f"ps.{r}.x"])


@pytest.mark.parametrize("s", [2, 3, 4])
def test_nearest_down_bit_exact(ops_golden, s):
    z = ops_golden
    y = orc.nearest_down(z[f"nd.{s}.x"], 1.0 / s)
    assert y.shape == z[f"nd.{s}.y"].shape
    assert np.array_equal(y, z[f"nd.{s}.y"])


def test_layernorm_cf(ops_golden):
    z = ops_golden
    _close(orc.layernorm_cf(z["ln.x"], z["ln.w"], z["ln.b"], 1e-6), z["ln.y"], 5e-6)


def test_esa_helpers(ops_golden):
    z = ops_golden
    assert np.array_equal(orc.max_pool2d(z["mp.x"], 7, 3), z["mp.y"])
    _close(orc.resize_bilinear(z["rb.x"], 23, 31), z["rb.y"], 2e-6)


def test_supersampler_forward(ss_golden):
    z, sd = ss_golden
    sr = orc.supersampler_forward(sd, z["lr_frames"], z["hr_buffers"])
    assert sr.shape == z["sr_frames"].shape
    _close(sr, z["sr_frames"], 2e-5)


def test_reprfn_deploy_forward(reprfn_golden):
    z, sd = reprfn_golden
    y = orc.reprfn_forward_deploy(sd, z["net.x"])
    assert y.shape == z["net.y"].shape
    _close(y, z["net.y"], 2e-5)


def test_supersampler_backward_golden(ss_golden):
    """The oracle's forward+backward composition reproduces the reference's autograd gradients
    (dL/d(sr) of L1 + 0.5*SSIM on the clipped output comes from torch's autograd of the loss mirror,
    checked here against the golden loss value)."""
    import torch
    from oracle.torch_checkers import LossGroup
    z, sd = ss_golden
    info = {}

    def dsr_fn(sr):
        t = torch.from_numpy(sr).requires_grad_(True)
        loss, ld = LossGroup(1.0, 0.5)({"hr_frames": torch.from_numpy(z["hr_frames"]), "sr_frames": torch.clip(t, 0.0, 10.0)})
        loss.backward()
        info.update(ld)
        return t.grad.numpy()

    sr, grads = orc.supersampler_forward_backward(sd, z["lr_frames"], z["hr_buffers"], dsr_fn)
    assert np.abs(sr - z["sr_frames"]).max() <= 1e-4
    assert abs(info["total"] - float(z["loss"])) <= 1e-5 and abs(info["ssim"] - float(z["loss_ssim"])) <= 1e-5
    for k in sd:
        ref = z["grad." + k]
        assert np.abs(grads[k] - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-30) + 1e-9, k


def test_loss_mirror_golden(ss_golden):
    """The torch formulation of LossGroup (the checker of the fused GPU loss) reproduces the loss values the
    reference computed on its own output."""
    import torch
    from oracle.torch_checkers import LossGroup
    z, _ = ss_golden
    loss, d = LossGroup(1.0, 0.5)({"hr_frames": torch.from_numpy(z["hr_frames"]),
                                   "sr_frames": torch.clip(torch.from_numpy(z["sr_frames"]), 0.0, 10.0)})
    assert abs(d["total"] - float(z["loss"])) <= 1e-6
    assert abs(d["l1"] - float(z["loss_l1"])) <= 1e-6 and abs(d["ssim"] - float(z["loss_ssim"])) <= 1e-6


@pytest.fixture(scope="module")
def temporal_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "temporal_ops.npz"))


def test_temporal_ops_oracle_golden(temporal_golden):
    """Operators of core/models/model_final.py: the oracle against outputs of the reference's own modules."""
    z = temporal_golden
    assert np.abs(orc.gradient_map(z["grad.x"]) - z["grad.y"]).max() <= 2e-6
    for name in ("t64_64", "t192_192"):
        y = orc.conv_transpose2x2s2(z[f"convT.{name}.x"], z[f"convT.{name}.w"], z[f"convT.{name}.b"], relu=True)
        assert np.abs(y - z[f"convT.{name}.y"]).max() <= 2e-5, name
    for key in [k[:-2] for k in z.files if k.startswith("tconv.") and k.endswith(".x")]:
        k, st, pd = (int(v) for v in z[key + ".geom"])
        y = orc.conv2d(z[key + ".x"], z[key + ".w"], z[key + ".b"], st, pd, act=1)
        assert np.abs(y - z[key + ".y"]).max() <= 5e-5, key
    assert np.abs(orc.resize_bilinear_align_corners(z["up2.x"], 10, 18) - z["up2.y"]).max() <= 2e-6


def test_layernorm_cf_backward_oracle_vs_torch_autograd():
    """tests/golden/ln_bwd.npz: torch autograd of the reference's channels-first LayerNorm formula."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ln_bwd.npz"))
    for tag in "abc":
        dx, dw, db = orc.layernorm_cf_bwd(z[f"{tag}.x"], z[f"{tag}.w"], z[f"{tag}.dy"])
        _close(dx, z[f"{tag}.dx"], 1e-5)
        _close(dw, z[f"{tag}.dw"], 1e-5 * max(1.0, float(np.abs(z[f"{tag}.dw"]).max())))
        _close(db, z[f"{tag}.db"], 1e-5 * max(1.0, float(np.abs(z[f"{tag}.db"]).max())))


def test_dilate_velocity_matches_the_reference_function():
    """core/tools/math.py:80-100 run on single samples (tests/golden/dilate_velocity.npz): ties, borders where the
    zero padding of F.unfold wins, a one-row frame, 3x3 and 5x5 windows -- bit-exact."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dilate_velocity.npz"))
    for tag in ("a", "b", "c"):
        got = orc.dilate_velocity(z[f"{tag}.motion"], z[f"{tag}.depth"], int(z[f"{tag}.k"]))
        assert np.array_equal(got, z[f"{tag}.out"]), tag
    # samples of a batch are independent
    m = np.concatenate([z["a.motion"], z["a.motion"][:, ::-1].copy()])
    d = np.concatenate([z["a.depth"], z["a.depth"][:, :, ::-1].copy()])
    both = orc.dilate_velocity(m, d, 3)
    assert np.array_equal(both[0:1], z["a.out"]) and np.array_equal(both[1:2], orc.dilate_velocity(m[1:2], d[1:2], 3))


def test_torch_operator_composition_reproduces_the_reference_forward(ss_golden):
    """bench.py times this composition of stock torch operators as the CPU baseline: it must be the reference's graph."""
    import torch
    from oracle.torch_checkers import supersampler_forward_torch
    z, sd = ss_golden
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        sr = supersampler_forward_torch(sd, torch.from_numpy(z["lr_frames"])[:, -1], torch.from_numpy(z["hr_buffers"])[:, -1])
    assert float((sr - torch.from_numpy(z["sr_frames"])).abs().max()) <= 1e-5
