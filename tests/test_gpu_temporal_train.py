"""Training graph of the temporal SuperSampler mirror against the reference's own autograd
(tests/golden/temporal_train.npz: core/models/model_final.py SuperSampler(3,10,4,1) in training mode with the weights of
temporal_model.npz on seeded inputs at LR 48 x 64 (tests/golden/seeded_inputs.py builds them on both sides), gradients
of every parameter under an enhance-weighted L1 term plus a dense linear term).  6.8 M gradients are not stored: the fixture holds, per tensor, the L2 norm, eight projections on seeded
+-1 vectors and, for tensors of at most 4,096 elements, the tensor itself.  The projections of the difference estimate
its norm (E[(v.e)^2] = |e|^2 for a +-1 vector v).

Bars.  The parity proof is test_temporal_gradients_against_the_bf16_storage_oracle below: the oracle
(oracle/temporal_emul.py, whose fp32 mode tests/test_temporal_emul_cpu.py pins to this fixture at 3e-4) rounds where the
GPU path stores bf16 and runs its backward on the GPU pass's own stored maps, and every one of the 113 tensors is held to
rel <= 2e-2, cosine >= 0.999.  The comparison with the fp32 reference itself stays as a guard; its bar follows the
number of pixels a gradient is summed over: the U-shaped main stage works at 1/2, 1/4
and 1/8 of the LR size, where this input leaves 24 x 32, 12 x 16 and 6 x 8 pixels, and a bf16 rounding or a ReLU that
flips at a value within rounding of zero moves such a sum by whole percents (at LR 16 x 24 the same tensors are 1.3-3
times further off, the large ones 1.5 times: the error falls with the pixel count, which a wrong formula's would not).
Per tensor: relative error <= 3e-2, <= 6e-2 at 1/2, <= 8e-2 at 1/4, <= 0.15 at 1/8 (the four one-channel gate layers: see
_bar); over all tensors together (error and gradient norms summed in squares) <= 1.5e-2; every norm within 10 %."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from seeded_inputs import temporal_train_inputs  # noqa: E402


def _bar(name: str) -> float:
    if any(k in name for k in (".conv3.", ".up3.", ".down3.")):
        return 0.15
    if any(k in name for k in (".w_self.", ".w_cross.")):
        # the four one-channel gate layers (gradient norms 1e-9 .. 1e-6 beside 1e-1 elsewhere: sums of cancelling terms).
        # In the bf16-storage oracle itself, moving 3 % of ONE feature map's elements by one bf16 step moves these
        # gradients by 6 - 25 % and the output layer's by 0.3 %: against the fp32 reference they are only guarded
        # against gross errors; their parity bar is the oracle test's 2e-2.
        return 0.5
    if any(k in name for k in (".conv2.", ".up2.", ".down2.")):
        return 8e-2
    if any(k in name for k in ("main_network.main_network.conv1.", "main_network.main_network.down1.", "main_network.main_network.up1.")):
        return 6e-2
    return 3e-2


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _model_and_inputs(h=48, w=64):
    from superresolution_amd.temporal import SuperSampler
    z = np.load(os.path.join(GOLD, "temporal_model.npz"))
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 4, 1)
    with torch.no_grad():
        for name, p in m.state_dict().items():
            if p.dim() > 1:
                p.copy_(_bf16_round(p))
            else:
                p.copy_(torch.from_numpy(z["small." + name]))
    args, target, G = temporal_train_inputs(h, w)
    return m.cuda(), [a.cuda() for a in args], target.cuda(), G.cuda()


def _tensor_errors(m, t):
    """[(relative error, name, reference norm, GPU norm)] of every parameter gradient of m against fixture t."""
    names = [str(n) for n in t["names"]]
    params = dict(m.named_parameters())
    assert names == list(params)
    rows = []
    for i, name in enumerate(names):
        g = params[name].grad
        assert g is not None and bool(torch.isfinite(g).all()), name
        gd = g.detach().double().reshape(-1).cpu()
        norm_ref = float(t["norm." + name][0])
        gen = torch.Generator().manual_seed(1000 + i)
        e2 = 0.0
        for k in range(8):
            v = torch.randint(0, 2, (gd.numel(),), generator=gen).double() * 2.0 - 1.0
            e2 += (float((gd * v).sum()) - float(t["proj." + name][k])) ** 2
        rel = (e2 / 8.0) ** 0.5 / max(norm_ref, 1e-12)
        if "grad." + name in t.files:
            ref = torch.from_numpy(t["grad." + name]).double().reshape(-1)
            rel = max(rel, float((gd - ref).norm()) / max(norm_ref, 1e-12))
        rows.append((rel, name, norm_ref, float(gd.norm())))
    return rows


def _loss(sr, ew, target, G):
    wmap = torch.nn.functional.interpolate(ew, size=sr.shape[-2:], mode="nearest")
    return ((1.0 + wmap) * (sr - target).abs()).mean() + (sr * G).mean()


def test_temporal_training_gradients_golden():
    t = np.load(os.path.join(GOLD, "temporal_train.npz"))
    m, args, target, G = _model_and_inputs()
    m.train()
    sr, ew = m(*args)
    assert sr.requires_grad and sr.shape == (1, 3, 192, 256)
    assert float(np.abs(ew.cpu().numpy() - t["enhance_weight"]).max()) <= 1e-6
    assert float(np.abs(sr.detach().cpu().numpy() - t["sr"].astype(np.float32)).max()) <= 2e-2
    loss = _loss(sr, ew, target, G)
    assert abs(float(loss.detach()) - float(t["loss"][0])) <= 2e-3
    loss.backward()
    worst, err2, ref2 = [], 0.0, 0.0
    for rel, name, norm_ref, norm_gpu in _tensor_errors(m, t):
        err2 += (rel * norm_ref) ** 2
        ref2 += norm_ref ** 2
        # (norms of gradients four and more orders of magnitude under the largest are sums of cancelling terms at bf16 storage:
        # they are held to the bf16-storage oracle below, not to this guard)
        if norm_ref >= 1e-5:
            assert abs(norm_gpu - norm_ref) <= 0.10 * norm_ref + 1e-12, (name, norm_gpu, norm_ref)
        worst.append((rel / _bar(name), rel, name, norm_ref))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:8]
    assert (err2 / ref2) ** 0.5 <= 1.5e-2, (err2 / ref2) ** 0.5


def _nchw(rec):
    """A recorded map of the GPU pass as the oracle's NCHW fp32 tensor."""
    t, c = rec
    if c is None:
        return t.float().cpu()
    return t[..., :c].permute(0, 3, 1, 2).float().cpu().contiguous()


def test_temporal_gradients_against_the_bf16_storage_oracle():
    """All 113 parameter gradients of the HIP training graph against oracle/temporal_emul.py in its bf16-storage mode,
    its backward run on the GPU pass's own stored maps (same ReLU decisions): rel <= 2e-2 and cosine >= 0.999 per
    tensor.  The forward maps themselves are checked against the UNFORCED emulation: within a few bf16 steps each."""
    from oracle import temporal_emul as TE
    from superresolution_amd.temporal import trace_maps
    from tests.test_temporal_emul_cpu import cpu_model, loss_fn
    m, args, target, G = _model_and_inputs()
    m.train()
    rec = {}
    with trace_maps(m, rec):
        sr, ew = m(*[a.clone() for a in args])
    _loss(sr, ew, target, G).backward()
    gpu = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    forced = {k: _nchw(v) for k, v in rec.items()}

    cm = cpu_model()
    cargs, ctarget, cG = temporal_train_inputs(48, 64)
    own = {}
    TE.forward(cm, cargs, emulate=True, trace=own)
    assert len(forced) >= 60 and set(forced) <= set(own), sorted(set(forced) - set(own))
    for k, v in forced.items():
        ref = own[k]
        assert v.shape == ref.shape, (k, v.shape, ref.shape)
        if k.startswith("pre."):
            assert float((v - ref).abs().max()) <= 1e-4 or float(((v - ref).abs() > 1e-4).float().mean()) <= 2e-3, k   # fp32 input-side tensors
            continue
        tol = 2.0 ** -6 * max(1.0, float(ref.abs().max()))      # four bf16 steps at the map's magnitude
        assert float((v - ref).abs().max()) <= tol, (k, float((v - ref).abs().max()), tol)

    sr_o, ew_o, loss_o, grads = TE.forward_backward(cm, cargs, loss_fn(ctarget, cG), emulate=True, forced=forced)
    assert float((sr.detach().cpu() - sr_o).abs().max()) == 0.0       # the last layer's stored map is forced
    worst = []
    for n, g in gpu.items():
        ref = grads[n]
        nr = float(ref.norm())
        rel = float((g - ref).norm()) / max(nr, 1e-30)
        cos = float((g * ref).sum()) / max(float(g.norm()) * nr, 1e-30)
        worst.append((rel, 1.0 - cos, n, nr))
    worst.sort(reverse=True)
    assert worst[0][0] <= 2e-2, worst[:6]
    assert max(w[1] for w in worst) <= 1e-3, sorted(worst, key=lambda w: -w[1])[:6]


def test_temporal_gradient_error_falls_with_the_pixel_count():
    """What separates rounding noise from a wrong formula: the same comparison at LR 16 x 24
    (tests/golden/temporal_train_small.npz), a twelfth of the pixels, leaves the gradients of the low-resolution stages
    (1/4 and 1/8 of the LR size: 4 x 6 and 2 x 3 pixels there) clearly further from the reference than at LR 48 x 64,
    while a systematic error would not care how many pixels a sum runs over."""
    deep = lambda name: any(k in name for k in (".conv2.", ".up2.", ".down2.", ".conv3.", ".up3.", ".down3."))
    means = {}
    for hw, fixture in (((48, 64), "temporal_train.npz"), ((16, 24), "temporal_train_small.npz")):
        t = np.load(os.path.join(GOLD, fixture))
        m, args, target, G = _model_and_inputs(*hw)
        m.train()
        sr, ew = m(*args)
        loss = _loss(sr, ew, target, G)
        assert abs(float(loss.detach()) - float(t["loss"][0])) <= 2e-3
        loss.backward()
        rels = [rel for rel, name, _, _ in _tensor_errors(m, t) if deep(name)]
        assert len(rels) >= 24
        means[hw] = float(np.mean(rels))
    assert means[(16, 24)] >= 1.3 * means[(48, 64)], means
    assert means[(48, 64)] <= 0.08 and means[(16, 24)] <= 0.2, means


def test_temporal_training_step_reduces_the_loss_and_repeats():
    """A few AdamW steps on one sample lower the loss; the first step's gradients repeat bit for bit."""
    def first_grads():
        m, args, target, G = _model_and_inputs()
        m.train()
        sr, ew = m(*[a.clone() for a in args])
        _loss(sr, ew, target, G).backward()
        return m, args, target, G, [p.grad.clone() for p in m.parameters()]

    m, args, target, G, g0 = first_grads()
    g1 = first_grads()[-1]
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        sr, ew = m(*[a.clone() for a in args])
        loss = _loss(sr, ew, target, G)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("shape", [(2, 6, 8, 32), (1, 12, 16, 64), (1, 5, 7, 16)])
def test_x2_align_corners_resize_backward_is_the_adjoint(shape):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) (core/models/model_final.py:281-289) in the
    training graph: forward and backward against torch's fp32 operator and its autograd on the same bf16 values."""
    from superresolution_amd import autograd_ops as A_
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h * 31 + w)
    x = (torch.rand(n, h, w, c, generator=g) - 0.5).to(torch.bfloat16).cuda().requires_grad_(True)
    dy = (torch.rand(n, 2 * h, 2 * w, c, generator=g) - 0.5).to(torch.bfloat16).cuda()
    y = A_.resize_bilinear(x, 2 * h, 2 * w, align_corners=True)
    y.backward(dy)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    assert float((y.detach().float().permute(0, 3, 1, 2) - yr.detach()).abs().max()) <= 2.0 ** -8
    assert float((x.grad.float().permute(0, 3, 1, 2) - xr.grad).abs().max()) <= 2.0 ** -7 * float(xr.grad.abs().max())


def test_chained_training_nodes_give_the_same_gradients_bit_for_bit(monkeypatch):
    """Runs of conv + ReLU layers are one autograd node in the training graph: each intermediate ReLU backward rides on the
    data-gradient launch behind it (mask epilogue of the implicit-GEMM / register-resident kernel) instead of a pass of
    its own.  A mask only zeroes, so forward and every one of the 113 gradients equal the layer-by-layer graph's exactly
    (which the bf16-storage oracle test above pins, in recorded passes)."""
    from superresolution_amd import temporal
    outs = []
    for chained in (True, False):
        monkeypatch.setattr(temporal, "_CHAIN_TRAIN", chained)
        m, args, target, G = _model_and_inputs()
        m.train()
        sr, ew = m(*[a.clone() for a in args])
        _loss(sr, ew, target, G).backward()
        outs.append((sr.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) == 113
    for name, g in outs[0][1].items():
        assert torch.equal(g, outs[1][1][name]), name
