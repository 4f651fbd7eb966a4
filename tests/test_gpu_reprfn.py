"""GPU parity of the RepRFN mirror and of the ESA helper kernels, against the reference-generated
goldens and the CPU oracle.  bf16 activations: max-abs-error <= 2e-2."""
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


def test_reprfn_deploy_forward_golden(reprfn_golden):
    from superresolution_amd.reprfn import RepRFN
    z, sd = reprfn_golden
    m = RepRFN(3, 3, 48, 4, deploy=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(dev(z["net.x"])).cpu().numpy()
    assert y.shape == z["net.y"].shape and y.dtype == np.float32
    assert np.abs(y - z["net.y"]).max() <= 2e-2
    assert np.abs(y - orc.reprfn_forward_deploy(sd, z["net.x"])).max() <= 2e-2


def test_reprfn_training_form_equals_deploy_form():
    """A training-form state dict gives the deploy result without switching (kernel re-parameterised on
    the fly), and switch_to_deploy() leaves the output unchanged."""
    from superresolution_amd.reprfn import RepRFN
    torch.manual_seed(5)
    m = RepRFN(3, 3, 48, 4, deploy=False).cuda().eval()
    x = torch.rand(2, 3, 40, 56, device="cuda")
    with torch.no_grad():
        y_train = m(x)
        for mod in m.modules():
            if hasattr(mod, "switch_to_deploy"):
                mod.switch_to_deploy()
        y_dep = m(x)
    assert y_train.shape == (2, 3, 160, 224)
    assert torch.equal(y_train, y_dep)
    # and against the oracle's deploy forward on the switched weights
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = orc.reprfn_forward_deploy(sd, x.cpu().numpy())
    assert np.abs(y_dep.cpu().numpy() - ref).max() <= 2e-2


@pytest.mark.parametrize("shape", [(2, 16, 30, 41), (1, 48, 7, 7), (1, 16, 19, 8)])
def test_max_pool_nhwc_bit_exact(F_, shape):
    rng = np.random.default_rng(shape[2])
    x = bf16r(rng.standard_normal(shape))
    y = F_.max_pool2d_nhwc(F_.nchw_f32_to_nhwc_bf16(dev(x)), 7, 3)
    got = F_.nhwc_bf16_to_nchw_f32(y, shape[1]).cpu().numpy()
    assert np.array_equal(got, orc.max_pool2d(x, 7, 3))


@pytest.mark.parametrize("cfg", [(3, 5, 30, 41), (1, 1, 9, 9), (8, 13, 16, 24), (6, 6, 6, 6)])
def test_resize_bilinear_nhwc_vs_oracle(F_, cfg):
    h, w, ho, wo = cfg
    rng = np.random.default_rng(h * 10 + wo)
    x = bf16r(rng.standard_normal((2, 16, h, w)))
    add = bf16r(rng.standard_normal((2, 16, ho, wo)))
    xa = F_.nchw_f32_to_nhwc_bf16(dev(x))
    got = F_.nhwc_bf16_to_nchw_f32(F_.resize_bilinear_nhwc(xa, ho, wo), 16).cpu().numpy()
    ref = orc.resize_bilinear(x, ho, wo)
    assert np.abs(got - ref).max() <= 2e-2
    got2 = F_.nhwc_bf16_to_nchw_f32(F_.resize_bilinear_nhwc(xa, ho, wo, addend=F_.nchw_f32_to_nhwc_bf16(dev(add))), 16).cpu().numpy()
    assert np.abs(got2 - (ref + add)).max() <= 4e-2


def test_sigmoid_gate(F_):
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((1, 5, 9, 48)).astype(np.float32)).to(torch.bfloat16).cuda()
    g = torch.from_numpy((rng.standard_normal((1, 5, 9, 48)) * 4).astype(np.float32)).to(torch.bfloat16).cuda()
    got = F_.sigmoid_gate(x, g).float().cpu().numpy()
    ref = x.float().cpu().numpy() / (1.0 + np.exp(-g.float().cpu().numpy()))
    assert np.abs(got - ref).max() <= 2e-2


@pytest.mark.parametrize("cfg", [(1, 48, 30, 41, 48), (2, 48, 64, 57, 64), (1, 16, 21, 33, 16), (1, 64, 40, 24, 64), (1, 32, 17, 90, 32)])
def test_esa_tail_one_pass_vs_oracle_and_layer_by_layer(F_, cfg):
    """x * sigmoid(conv4(resize(c3) + conv_f(c1))) (core/arch/RepRFN.py:303-308) as one launch: against the oracle's
    operators on the same bf16 inputs and bf16-rounded weights (<= 2e-2), and against the four launches it replaces
    (same rounding points; the sums run in another order: a bf16 step of the gate at most)."""
    n, cf_, h, w, cs = cfg
    rng = np.random.default_rng(h * 100 + w)
    h3, w3 = max(1, ((h - 3) // 2 + 1 - 7) // 3 + 1), max(1, ((w - 3) // 2 + 1 - 7) // 3 + 1)
    x = bf16r(rng.standard_normal((n, cf_, h, w)))
    c1 = bf16r(rng.standard_normal((n, 16, h, w)))
    c3 = bf16r(rng.standard_normal((n, 16, h3, w3)))
    wf, w4 = bf16r(rng.standard_normal((16, 16, 1, 1)) * 0.3), bf16r(rng.standard_normal((cf_, 16, 1, 1)) * 0.3)
    bf_, b4 = rng.standard_normal(16).astype(np.float32) * 0.1, rng.standard_normal(cf_).astype(np.float32) * 0.1
    xa = F_.nchw_f32_to_nhwc_bf16(dev(x), cpad=cs)
    c1a, c3a = F_.nchw_f32_to_nhwc_bf16(dev(c1)), F_.nchw_f32_to_nhwc_bf16(dev(c3))
    got = F_.esa_tail(xa, c1a, c3a, dev(wf), dev(bf_), dev(w4), dev(b4), cf_)
    assert got.shape == xa.shape and bool((got[..., cf_:] == 0).all())
    got_n = F_.nhwc_bf16_to_nchw_f32(got, cf_).cpu().numpy()
    # oracle, with the maps the layer-by-layer form stores rounded to bf16
    cfo = bf16r(orc.conv2d(c1, wf, bf_, 1, 0))
    so = bf16r(orc.resize_bilinear(c3, h, w) + cfo)
    c4o = bf16r(orc.conv2d(so, w4, b4, 1, 0))
    ref = x * (np.float32(1.0) / (np.float32(1.0) + np.exp(-c4o, dtype=np.float32)))
    assert np.abs(got_n - ref).max() <= 2e-2
    # the four launches
    cfl = F_.conv2d_auto(c1a, dev(wf), dev(bf_), stride=1, padding=0)
    sl = F_.resize_bilinear_nhwc(c3a, h, w, addend=cfl)
    c4l = F_.conv2d_auto(sl, dev(w4), dev(b4), stride=1, padding=0)
    if c4l.shape[-1] != xa.shape[-1]:
        c4l = torch.nn.functional.pad(c4l, (0, xa.shape[-1] - c4l.shape[-1])) if c4l.shape[-1] < xa.shape[-1] else c4l[..., :xa.shape[-1]].contiguous()
    lay = F_.nhwc_bf16_to_nchw_f32(F_.sigmoid_gate(xa, c4l), cf_).cpu().numpy()
    assert np.abs(got_n - lay).max() <= 2e-2 * max(1.0, float(np.abs(x).max()))
    assert np.mean(got_n != lay) <= 0.05
