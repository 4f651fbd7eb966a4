"""GPU parity of the warp / coordinate mirrors (reference names and defaults) against the goldens the
reference produced and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("align", [True, False])
def test_backward_warping_golden(ops_golden, mode, align):
    from superresolution_amd.tools_math import backward_warping
    z = ops_golden
    y = backward_warping(dev(z["warp.frame"]), dev(z["warp.motion"]), mode, align).cpu().numpy()
    ref = z[f"warp.{mode}.{int(align)}"]
    if mode == "nearest":
        assert np.array_equal(y, ref)
    else:
        assert np.abs(y - ref).max() <= 2e-6


def test_backward_warping_defaults_are_the_references(ops_golden):
    from superresolution_amd.tools_math import backward_warping
    z = ops_golden
    y = backward_warping(dev(z["warp.frame"]), dev(z["warp.motion"])).cpu().numpy()   # nearest, align_corners=True
    assert np.array_equal(y, z["warp.nearest.1"])
    with pytest.raises(ValueError):
        backward_warping(dev(z["warp.frame"]), dev(z["warp.motion"]), "bicubic")


def test_backward_warping_autograd_golden(ops_golden):
    from superresolution_amd.tools_math import backward_warping
    z = ops_golden
    fr = dev(z["warp.frame"]).requires_grad_(True)
    mo = dev(z["warp.motion"]).requires_grad_(True)
    y = backward_warping(fr, mo, "bilinear", True)
    y.backward(dev(z["warpbwd.dy"]))
    assert np.abs(fr.grad.cpu().numpy() - z["warpbwd.dframe"]).max() <= 1e-5
    assert np.abs(mo.grad.cpu().numpy() - z["warpbwd.dmotion"]).max() <= 1e-4


def test_backward_warping_history_vs_oracle():
    """core/models/model_final.py:17-37: accumulate the motion over the history, then warp frames and buffers."""
    from superresolution_amd.tools_math import backward_warping_history
    rng = np.random.default_rng(11)
    b, f, h, w = 2, 3, 18, 30
    mv = (rng.standard_normal((b, f, 2, h, w)) * 2).astype(np.float32)
    frames = rng.random((b, f, 3, h, w), dtype=np.float32)
    bufs = rng.random((b, f, 10, h, w), dtype=np.float32)
    acc = mv.copy()
    for i in range(1, f):
        acc[:, i] = orc.backward_warp(acc[:, i], acc[:, i - 1], "nearest", True) + acc[:, i - 1]
    wf_ref = orc.backward_warp(frames.reshape(b * f, 3, h, w), acc.reshape(b * f, 2, h, w), "bilinear", True)
    wb_ref = orc.backward_warp(bufs.reshape(b * f, 10, h, w), acc.reshape(b * f, 2, h, w), "bilinear", True)
    wf, wb, mva = backward_warping_history(dev(mv), dev(frames), dev(bufs))
    assert np.abs(mva.cpu().numpy() - acc).max() <= 1e-5
    assert np.abs(wf.cpu().numpy().reshape(wf_ref.shape) - wf_ref).max() <= 1e-4
    assert np.abs(wb.cpu().numpy().reshape(wb_ref.shape) - wb_ref).max() <= 1e-4


def test_make_coord_mirror(ops_golden):
    from superresolution_amd.tools_math import make_coord
    c = make_coord((5, 7))
    assert c.shape == (1, 2, 5, 7) and c.dtype == torch.float32
    assert np.array_equal(c.cpu().numpy(), orc.make_coord(5, 7))
    assert make_coord((5, 7), acquire_batch_dim=False).shape == (2, 5, 7)
    with pytest.raises(NotImplementedError):
        make_coord((5, 7), ranges=((0, 1), (0, 1)))
