"""Generate the golden fixtures in tests/golden/ by importing the reference on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference modules are imported unmodified; third-party modules that only do
I/O (torchvision, cv2, omegaconf -- none on the arithmetic path) are replaced by
empty stubs so the import succeeds.  Outputs are small .npz files of inputs and
expected outputs; no reference source text is stored.

Known reference defect handled here: core/tools/math.py:16-17 stacks the warp
grid on dim=-1 and then permutes it again, which only runs when H == 2.  The
warp goldens therefore call F.grid_sample with the grid the function evidently
means to build (gx, gy exactly as math.py:12-15, stacked on the last dim).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("SR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    for name in ["torchvision", "torchvision.utils", "torchvision.models",
                 "torchvision.models.feature_extraction", "cv2", "omegaconf"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision.models"].feature_extraction = sys.modules["torchvision.models.feature_extraction"]
    sys.modules["omegaconf"].OmegaConf = object
    sys.path.insert(0, REF)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def to_u16(t: torch.Tensor) -> np.ndarray:
    """bf16-representable fp32 tensor -> uint16 bit pattern (halves fixture size)."""
    return (t.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)


def rnd(gen, *shape, lo=-1.0, hi=1.0):
    return bf16_round(torch.rand(*shape, generator=gen) * (hi - lo) + lo)


def gen_ops():
    from core.tools.math import make_coord
    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- convolutions (nn.Conv2d call sites of model.py / RepRFN.py / model_final.py)
    conv_cases = [
        # name, N, Cin, Cout, H, W, kh, kw, stride, pad
        ("c64_64", 1, 64, 64, 10, 37, 3, 3, 1, 1),
        ("c10_64", 1, 10, 64, 9, 21, 3, 3, 1, 1),
        ("c96_64", 1, 96, 64, 7, 19, 3, 3, 1, 1),
        ("c64_3", 2, 64, 3, 8, 13, 3, 3, 1, 1),
        ("c3_32", 1, 3, 32, 12, 20, 3, 3, 1, 1),
        ("c32_32", 1, 32, 32, 12, 20, 3, 3, 1, 1),
        ("c48_48", 1, 48, 48, 9, 14, 3, 3, 1, 1),
        ("c48_16_1x1", 1, 48, 16, 9, 14, 1, 1, 1, 0),
        ("c16_16_s2p0", 1, 16, 16, 15, 17, 3, 3, 2, 0),
        ("c32_48_s2p1", 1, 32, 48, 12, 16, 3, 3, 2, 1),
        ("c48_48_3x1", 1, 48, 48, 9, 14, 3, 1, 1, (1, 0)),
        ("c48_48_1x3", 1, 48, 48, 9, 14, 1, 3, 1, (0, 1)),
    ]
    names = []
    for (name, N, Cin, Cout, H, W, kh, kw, s, p) in conv_cases:
        x = rnd(g, N, Cin, H, W)
        w = bf16_round(torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5)
        b = bf16_round(torch.randn(Cout, generator=g) * 0.1)
        y = F.conv2d(x, w, b, stride=s, padding=p)
        out[f"conv.{name}.x"] = x.numpy(); out[f"conv.{name}.w"] = w.numpy()
        out[f"conv.{name}.b"] = b.numpy(); out[f"conv.{name}.y"] = y.numpy()
        pp = p if isinstance(p, tuple) else (p, p)
        out[f"conv.{name}.cfg"] = np.array([s, pp[0], pp[1]], np.int32)
        names.append(name)
    out["conv.names"] = np.array(names)

    # ---- conv backward (autograd of nn.Conv2d): 64->64 and 10->64
    for (name, Cin, Cout, H, W) in [("b64_64", 64, 64, 6, 9), ("b10_64", 10, 64, 5, 8), ("b64_3", 64, 3, 6, 7)]:
        x = rnd(g, 2, Cin, H, W).requires_grad_(True)
        w = bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).requires_grad_(True)
        b = torch.zeros(Cout, requires_grad=True)
        dy = rnd(g, 2, Cout, H, W)
        y = F.conv2d(x, w, b, padding=1)
        y.backward(dy)
        for k, v in (("x", x), ("w", w), ("dy", dy), ("dx", x.grad), ("dw", w.grad), ("db", b.grad)):
            out[f"convbwd.{name}.{k}"] = v.detach().numpy()

    # ---- make_coord (core/tools/math.py:107-130)
    for (h, w) in [(3, 4), (12, 20), (48, 80), (7, 5), (270, 480), (1080, 1920), (2160, 3840)]:
        c = make_coord((h, w))
        # store only the two 1-D profiles: the field is separable
        out[f"coord.{h}x{w}.x"] = c[0, 0, 0, :].numpy().copy()
        out[f"coord.{h}x{w}.y"] = c[0, 1, :, 0].numpy().copy()

    # ---- a-6: grid_sample(fmap, make_coord(th,tw)) bilinear zeros align_corners=False (model.py:153,162)
    for (name, C, sh, sw, th, tw) in [("s4", 32, 5, 7, 20, 28), ("s2", 32, 6, 5, 12, 10), ("s3", 8, 4, 6, 12, 18)]:
        x = rnd(g, 2, C, sh, sw)
        coord = make_coord((th, tw)).expand(2, 2, th, tw)
        y = F.grid_sample(x, coord.permute(0, 2, 3, 1), mode="bilinear", align_corners=False)
        out[f"up.{name}.x"] = x.numpy(); out[f"up.{name}.y"] = y.numpy()
    # backward of the same op
    x = rnd(g, 1, 32, 5, 7).requires_grad_(True)
    coord = make_coord((20, 28)).expand(1, 2, 20, 28)
    y = F.grid_sample(x, coord.permute(0, 2, 3, 1), mode="bilinear", align_corners=False)
    dy = rnd(g, 1, 32, 20, 28)
    y.backward(dy)
    out["upbwd.x"] = x.detach().numpy(); out["upbwd.dy"] = dy.numpy(); out["upbwd.dx"] = x.grad.numpy()

    # ---- a-7: relative coordinates (code_test.py:193-202; lte.py:164-171)
    for (sh, sw, s) in [(3, 4, 16), (6, 5, 2), (5, 7, 3), (5, 7, 4), (270, 480, 4)]:
        th, tw = sh * s, sw * s
        hr = make_coord((th, tw)).expand(1, 2, th, tw)
        lr = make_coord((sh, sw)).expand(1, 2, sh, sw)
        cs = F.grid_sample(lr, hr.permute(0, 2, 3, 1), mode="nearest", align_corners=False)
        rel = hr - cs
        rel[:, 0] *= sw
        rel[:, 1] *= sh
        if th * tw > 10000:  # separable: keep profiles only
            out[f"rel.{sh}x{sw}x{s}.x"] = rel[0, 0, 0, :].numpy().copy()
            out[f"rel.{sh}x{sw}x{s}.y"] = rel[0, 1, :, 0].numpy().copy()
        else:
            out[f"rel.{sh}x{sw}x{s}"] = rel.numpy()

    # ---- a-9: backward_warping semantics (math.py:10-17 with the evident grid layout)
    def warp(frame, motion, mode, align):
        h, w = frame.shape[-2:]
        gx = torch.arange(w).view(1, w).expand(h, -1).float()
        gy = torch.arange(h).view(h, 1).expand(-1, w).float()
        gx = ((gx[None, ...] + motion[:, 0]) / (w - 1)) * 2.0 - 1.0
        gy = ((gy[None, ...] + motion[:, 1]) / (h - 1)) * 2.0 - 1.0
        grid = torch.stack([gx, gy], dim=-1)
        return F.grid_sample(frame, grid, mode, "zeros", align)

    frame = rnd(g, 2, 13, 11, 17)
    motion = torch.randn(2, 2, 11, 17, generator=g) * 2.5
    motion[0, :, 0, :] = torch.tensor([-30.0, 40.0]).view(2, 1)  # out of frame
    motion[1, :, 3, 4] = torch.tensor([0.5, -0.5])               # exact half-pixel
    motion[1, :, 5, :] = 0.0                                     # identity row
    motion = bf16_round(motion)
    out["warp.frame"] = frame.numpy(); out["warp.motion"] = motion.numpy()
    for mode in ("bilinear", "nearest"):
        for align in (True, False):
            out[f"warp.{mode}.{int(align)}"] = warp(frame, motion, mode, align).numpy()
    # MV accumulation step (model_final.py:22-24): mv1 = warp(mv1, mv0, bilinear, True) + mv0
    mv0, mv1 = motion[:1], motion[1:]
    out["warp.accum"] = (warp(mv1, mv0, "bilinear", True) + mv0).numpy()
    # backward (bilinear, align_corners=True) w.r.t. frame and motion
    fr = frame.clone().requires_grad_(True); mo = motion.clone().requires_grad_(True)
    yw = warp(fr, mo, "bilinear", True)
    dyw = rnd(g, *yw.shape)
    yw.backward(dyw)
    out["warpbwd.dy"] = dyw.numpy(); out["warpbwd.dframe"] = fr.grad.numpy(); out["warpbwd.dmotion"] = mo.grad.numpy()

    # ---- a-10: pixel shuffle / unshuffle (RepRFN.py:272; model_final.py:234,449)
    for r in (2, 4):
        x = torch.arange(2 * 3 * r * r * 5 * 7, dtype=torch.float32).view(2, 3 * r * r, 5, 7)
        out[f"ps.{r}.x"] = x.numpy(); out[f"ps.{r}.y"] = F.pixel_shuffle(x, r).numpy()
        z = torch.arange(2 * 10 * (3 * r) * (4 * r), dtype=torch.float32).view(2, 10, 3 * r, 4 * r)
        out[f"pu.{r}.x"] = z.numpy(); out[f"pu.{r}.y"] = F.pixel_unshuffle(z, r).numpy()

    # ---- a-11: nearest down-sampling (trainer.py:92-95)
    for s in (2, 3, 4):
        x = torch.arange(2 * 3 * 24 * 36, dtype=torch.float32).view(2, 3, 24, 36)
        y = F.interpolate(x, scale_factor=1.0 / s, mode="nearest")
        out[f"nd.{s}.x"] = x.numpy(); out[f"nd.{s}.y"] = y.numpy()

    # ---- a-8: channels-first LayerNorm (CAMixerSR.py:32-37; formula evaluated with torch ops,
    #      the class itself cannot be imported: its module needs basicsr)
    x = rnd(g, 2, 60, 6, 9, lo=-2.0, hi=2.0)
    wt = bf16_round(torch.rand(60, generator=g) + 0.5); bs = bf16_round(torch.randn(60, generator=g) * 0.1)
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    y = (x - u) / torch.sqrt(s + 1e-6)
    y = wt[:, None, None] * y + bs[:, None, None]
    out["ln.x"] = x.numpy(); out["ln.w"] = wt.numpy(); out["ln.b"] = bs.numpy(); out["ln.y"] = y.numpy()

    # ---- ESA helpers (RepRFN.py:298,303)
    x = rnd(g, 1, 16, 23, 29)
    out["mp.x"] = x.numpy(); out["mp.y"] = F.max_pool2d(x, kernel_size=7, stride=3).numpy()
    x = rnd(g, 1, 16, 5, 7)
    out["rb.x"] = x.numpy(); out["rb.y"] = F.interpolate(x, (23, 31), mode="bilinear", align_corners=False).numpy()

    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def gen_layernorm_bwd():
    """a-8 / a-13: autograd of the channels-first LayerNorm formula (CAMixerSR.py:32-37, evaluated with torch ops:
    the class's module needs basicsr) -> tests/golden/ln_bwd.npz."""
    g = torch.Generator().manual_seed(808)
    out = {}
    for tag, shape in (("a", (2, 60, 6, 9)), ("b", (1, 16, 5, 3)), ("c", (3, 7, 2, 11))):
        x = rnd(g, *shape, lo=-2.0, hi=2.0).requires_grad_(True)
        wt = (torch.rand(shape[1], generator=g) + 0.5).requires_grad_(True)
        bs = (torch.randn(shape[1], generator=g) * 0.1).requires_grad_(True)
        dy = rnd(g, *shape)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        y = (x - u) / torch.sqrt(s + 1e-6)
        y = wt[:, None, None] * y + bs[:, None, None]
        y.backward(dy)
        for k, v in (("x", x), ("w", wt), ("b", bs), ("dy", dy), ("y", y), ("dx", x.grad), ("dw", wt.grad), ("db", bs.grad)):
            out[f"{tag}.{k}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ln_bwd.npz"), **out)


def gen_supersampler():
    from core.models.model import SuperSampler, LossGroup
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 1)
    # digest of the seed-42 initialisation (first 8 values + fp64 sum of every parameter)
    init = {name: np.concatenate([p.detach().numpy().ravel()[:8],
                                  np.array([p.detach().double().sum().item()], np.float32)]).astype(np.float32)
            for name, p in m.named_parameters()}
    # bf16-representable weights and small random biases (the zero-bias init would hide bias bugs)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("bias"):
                p.copy_(bf16_round(torch.randn(p.shape, generator=g) * 0.05))
            else:
                p.copy_(bf16_round(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}

    B, L, sh, sw, s = 1, 2, 12, 20, 4
    lr = rnd(g, B, L, 3, sh, sw, lo=0.0, hi=1.0)
    hb = rnd(g, B, L, 10, sh * s, sw * s, lo=0.0, hi=1.0)
    hr = rnd(g, B, 3, sh * s, sw * s, lo=0.0, hi=1.0)
    batch = {"lr_frames": lr.clone(), "hr_buffers": hb.clone(), "hr_frames": hr, "scale": 4.0}
    outb = m(batch)
    sr = outb["sr_frames"]
    loss_group = LossGroup(1.0, 0.5)
    loss, loss_dict = loss_group({"hr_frames": hr, "sr_frames": torch.clip(sr, 0.0, 10.0)})
    loss.backward()

    out = {"lr_frames": lr.numpy(), "hr_buffers": hb.numpy(), "hr_frames": hr.numpy(),
           "sr_frames": sr.detach().numpy(), "loss": np.float32(loss.item()),
           "loss_l1": np.float32(loss_dict["l1"]), "loss_ssim": np.float32(loss_dict["ssim"])}
    for k, v in init.items():
        out["init." + k] = v
    out["keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        out["sd." + k] = to_u16(v)
    for name, p in m.named_parameters():
        gr = p.grad.detach()
        out["grad." + name] = gr.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "supersampler.npz"), **out)
    print("supersampler.npz: loss", loss.item(), loss_dict, "sr mean", sr.abs().mean().item())

    # one AdamW step as core/trainer.py:158-161 with configs/trainer/4x_H2.yaml:12-17 hyper-parameters
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    opt.step()
    step = {}
    for name in ["frame_extractor.blks.0.0.weight", "main.conv_main.3.blks.2.weight", "final_conv.2.bias",
                 "final_conv.2.weight"]:
        step["after." + name] = dict(m.named_parameters())[name].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "supersampler_step.npz"), **step)


def gen_reprfn():
    from core.arch.RepRFN import RepRFN, RepBlock
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(11)
    # (1) re-parameterisation of one RepBlock (RepRFN.py:193-241, SeqConv3x3.rep_params :132-146)
    blk = RepBlock(48, 48, "lrelu", deploy=False)
    with torch.no_grad():
        for p in blk.parameters():
            if p.requires_grad:
                p.copy_(bf16_round(p))
    x = rnd(g, 1, 48, 9, 14)
    y_train = blk(x)
    k, b = blk.get_equivalent_kernel_bias()
    out = {"blk.x": x.numpy(), "blk.y": y_train.detach().numpy(), "blk.k": k.detach().numpy(), "blk.b": b.detach().numpy()}
    out["blk.keys"] = np.array(list(blk.state_dict().keys()))
    for kk, v in blk.state_dict().items():
        out["blk.sd." + kk] = v.detach().numpy()

    # (2) full network, deploy form, bf16-representable weights
    m = RepRFN(3, 3, 48, 4, deploy=False)
    for mod in m.modules():
        if hasattr(mod, "switch_to_deploy"):
            mod.switch_to_deploy()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(bf16_round(p))
    m.eval()
    xin = rnd(g, 1, 3, 16, 24, lo=0.0, hi=1.0)
    with torch.no_grad():
        y = m(xin)
    out["net.x"] = xin.numpy(); out["net.y"] = y.numpy()
    out["net.keys"] = np.array(list(m.state_dict().keys()))
    for kk, v in m.state_dict().items():
        out["net.sd." + kk] = to_u16(v)
    np.savez_compressed(os.path.join(OUT, "reprfn.npz"), **out)
    print("reprfn.npz: y mean", y.abs().mean().item(), "params", sum(p.numel() for p in m.parameters()))


def gen_reprfn_train():
    """RepRFN in its TRAINING form (nine branches per block, RepRFN.py:183-191) run by the reference itself:
    forward and the gradient of sum(y * r) w.r.t. every parameter -> tests/golden/reprfn_train.npz.  The weights are
    not stored: seed 21 + bf16 rounding reproduces them (the construction order is identical); gradients are stored
    as fp16 of grad / max|grad| plus the scale."""
    from core.arch.RepRFN import RepRFN
    torch.manual_seed(21)
    m = RepRFN(3, 3, 48, 4, deploy=False)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.requires_grad:
                p_.copy_(bf16_round(p_))
    m.train()
    g = torch.Generator().manual_seed(22)
    x = bf16_round(rnd(g, 2, 3, 20, 28, lo=0.0, hi=1.0))
    r = rnd(g, 2, 3, 80, 112)
    y = m(x)
    (y * r).sum().backward()
    out = {"x": x.numpy(), "r": r.numpy(), "y": y.detach().numpy()}
    names = []
    for name, p_ in m.named_parameters():
        if not p_.requires_grad:
            continue
        names.append(name)
        gr = p_.grad.detach()
        sc = float(gr.abs().max())
        out["gscale." + name] = np.float32(sc)
        out["g." + name] = (gr / max(sc, 1e-30)).numpy().astype(np.float16)
        out["digest." + name] = np.array([p_.detach().double().sum().item(), p_.detach().double().abs().sum().item()])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "reprfn_train.npz"), **out)
    print("reprfn_train.npz:", len(names), "parameter gradients, y mean", float(y.abs().mean()))


def gen_reprfn_init():
    """Digest of the seed-42 initialisation of RepRFN in its training form (first 8 values + fp64 sum
    of every state-dict entry) and the key lists of both forms."""
    from core.arch.RepRFN import RepRFN
    torch.manual_seed(42)
    m = RepRFN(3, 3, 48, 4, deploy=False)
    out = {"train.keys": np.array(list(m.state_dict().keys()))}
    for name, p in m.state_dict().items():
        v = p.detach().reshape(-1)
        out["init." + name] = np.concatenate([v[:8].numpy(), np.array([v.double().sum().item()], np.float32)]).astype(np.float32)
        out["shape." + name] = np.array(p.shape, np.int64)
    d = RepRFN(3, 3, 48, 4, deploy=True)
    out["deploy.keys"] = np.array(list(d.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "reprfn_init.npz"), **out)
    print("reprfn_init.npz:", len(out["train.keys"]), "train keys,", len(out["deploy.keys"]), "deploy keys")


def gen_temporal_ops():
    """Operators of the temporal variant (core/models/model_final.py) at small shapes, run on the
    reference's own module classes: GradientMap, TransConvActivation, ConvActivation (1x1, stride 2),
    nn.Upsample(x2, bilinear, align_corners=True)."""
    torch.cuda.current_device = lambda: "cpu"   # core/tools/color.py builds constants on "the current device"
    from core.models.model_final import ConvActivation, GradientMap, TransConvActivation
    g = torch.Generator().manual_seed(23)
    out = {}
    x = rnd(g, 2, 6, 13, 18, lo=0.0, hi=1.0)
    out["grad.x"] = x.numpy(); out["grad.y"] = GradientMap(6)(x).numpy()
    torch.manual_seed(5)
    for name, (ci, co) in {"t64_64": (64, 64), "t192_192": (192, 192)}.items():
        m = TransConvActivation(ci, co, 2, 2)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(bf16_round(p_ if p_.dim() > 1 else torch.randn(p_.shape, generator=g) * 0.05))
        xi = bf16_round(rnd(g, 1, ci, 5, 7))
        out[f"convT.{name}.x"] = xi.numpy(); out[f"convT.{name}.w"] = m[0].weight.detach().numpy()
        out[f"convT.{name}.b"] = m[0].bias.detach().numpy(); out[f"convT.{name}.y"] = m(xi).detach().numpy()
    for name, (ci, co, k, st, pd) in {"s2_32_48": (32, 48, 3, 2, 1), "s2_192_256": (192, 256, 3, 2, 1),
                                        "p1_160_160": (160, 160, 1, 1, 0), "p1_112_64": (112, 64, 1, 1, 0),
                                        "c3_160_128": (160, 128, 3, 1, 1), "c3_24_64": (24, 64, 3, 1, 1)}.items():
        m = ConvActivation(ci, co, k, st, pd)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(bf16_round(p_ if p_.dim() > 1 else torch.randn(p_.shape, generator=g) * 0.05))
        xi = bf16_round(rnd(g, 1, ci, 9, 12))
        out[f"tconv.{name}.x"] = xi.numpy(); out[f"tconv.{name}.w"] = m[0].weight.detach().numpy()
        out[f"tconv.{name}.b"] = m[0].bias.detach().numpy(); out[f"tconv.{name}.y"] = m(xi).detach().numpy()
        out[f"tconv.{name}.geom"] = np.array([k, st, pd], np.int64)
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    xu = bf16_round(rnd(g, 2, 16, 5, 9))
    out["up2.x"] = xu.numpy(); out["up2.y"] = up(xu).numpy()
    np.savez_compressed(os.path.join(OUT, "temporal_ops.npz"), **out)
    print("temporal_ops.npz:", len(out), "arrays")


def gen_temporal_masks():
    """Disocclusion logic of the temporal variant, run on the reference's own functions
    (core/models/model_final.py:40-139) and composed as its forward does (:472-486, :501-504)
    -> tests/golden/temporal_masks.npz."""
    torch.cuda.current_device = lambda: "cpu"
    from core.models import model_final as mf
    g = torch.Generator().manual_seed(77)
    out = {}
    for tag, (b, f, h, w) in {"a": (2, 2, 13, 17), "b": (1, 1, 6, 5), "c": (1, 3, 9, 22)}.items():
        # spatially smooth content (a low-resolution random field, enlarged) plus fine noise: neighbouring pixels
        # of a rendered frame are similar, and the masks compare against neighbourhoods
        smooth = lambda c: F.interpolate(torch.rand(b, c, 3, 4, generator=g), size=(h, w), mode="bilinear", align_corners=True)
        lr_frame = (smooth(3) + (torch.rand(b, 3, h, w, generator=g) - 0.5) * 0.05).clamp(0.0, 1.0)
        lr_buffer = (smooth(10) + (torch.rand(b, 10, h, w, generator=g) - 0.5) * 0.05).clamp(0.0, 1.0)
        lr_buffer[:, 9] *= 4.0
        # history = current + small noise, with patches that deviate strongly (disoccluded regions)
        noise = lambda *shape: (torch.rand(*shape, generator=g) - 0.5) * 0.08
        big = lambda *shape: (torch.rand(*shape, generator=g) < 0.3).float() * (torch.rand(*shape, generator=g) - 0.5) * 1.6
        warped_frames = (lr_frame[:, None] + noise(b, f, 3, h, w) + big(b, f, 1, h, w)).clamp(0.0, 1.0)
        warped_buffer = lr_buffer[:, None] + noise(b, f, 10, h, w) + big(b, f, 10, h, w)
        warped_buffer[:, :, :9] = warped_buffer[:, :, :9].clamp(0.0, 1.0)
        warped_buffer[:, :, 9] = warped_buffer[:, :, 9].clamp(0.0, 4.0)
        # a few exact ties and exact zeros of the normal dot product
        warped_frames[:, :, :, 0, 0] = lr_frame[:, None, :, 0, 0]
        lr_buffer[:, 0:3, 1, 1] = 0.5
        current_buffer = lr_buffer.view(b, 1, -1, h, w).expand_as(warped_buffer)
        m_frame = mf.compute_mask_frame(lr_frame, warped_frames, ksize=5)
        m_color = mf.compute_mask_basecolor(current_buffer[:, :, 6:9], warped_buffer[:, :, 6:9])
        m_depth = mf.compute_mask_depth(current_buffer[:, :, -1:], warped_buffer[:, :, -1:])
        mask = m_frame | m_color | m_depth
        marked_frames = torch.where(mask, 0, warped_frames).view(b * f, -1, h, w)
        marked_buffer = torch.where(mask, 0, warped_buffer)
        aux = torch.concat((current_buffer[:, :, 0:3], current_buffer[:, :, 6:10], marked_buffer[:, :, 0:3],
                            marked_buffer[:, :, 6:10]), dim=2).reshape(b * f, -1, h, w)
        m_n = mf.compute_mask_normal(current_buffer[:, :, 0:3], warped_buffer[:, :, 0:3])
        m_r = mf.compute_mask_roughness(current_buffer[:, :, 5:6], warped_buffer[:, :, 5:6], mark_min_value=0.02)
        m_m = mf.compute_mask_metallic(current_buffer[:, :, 3:4], warped_buffer[:, :, 3:4], mark_max_value=0.95)
        ew = torch.concat([m_n, m_r, m_m], dim=1).float().mean(dim=2, keepdim=True).mean(dim=1)
        depth = rnd(g, b, 1, h, w, lo=0.0, hi=12.0)
        for k, v in (("lr_frame", lr_frame), ("lr_buffer", lr_buffer), ("warped_frames", warped_frames),
                     ("warped_buffer", warped_buffer), ("mask_frame", m_frame), ("mask_color", m_color),
                     ("mask_depth", m_depth), ("mask", mask), ("marked_frames", marked_frames), ("aux", aux),
                     ("mask_normal", m_n), ("mask_rough", m_r), ("mask_metal", m_m), ("enhance_weight", ew),
                     ("depth", depth), ("depth_clamped", mf.clamp_depth([depth.clone()])[0])):
            out[f"{tag}.{k}"] = v.numpy()
        print(tag, "mask fraction", float(mask.float().mean()), "frame", float(m_frame.float().mean()),
              "color", float(m_color.float().mean()), "depth", float(m_depth.float().mean()), "ew mean", float(ew.mean()))
    np.savez_compressed(os.path.join(OUT, "temporal_masks.npz"), **out)


def gen_dilate_velocity():
    """core/tools/math.py:80-100 run as it stands on single-sample inputs (for B = 1 its advanced indexing yields
    (1,1,1,h,w,2): the gathered motion of every pixel) -> tests/golden/dilate_velocity.npz."""
    from core.tools.math import dilate_velocity
    g = torch.Generator().manual_seed(123)
    out = {}
    for tag, (h, w, k) in {"a": (9, 13, 3), "b": (6, 5, 5), "c": (1, 7, 3)}.items():
        motion = torch.randn(1, h, w, 2, generator=g)
        # depth well above zero in the interior and quantised, so that ties occur; the zero padding of F.unfold wins at
        # the border, as in the reference
        depth = (torch.rand(1, 1, h, w, generator=g) * 6).round() / 2 + 0.5
        ref = dilate_velocity(motion, depth, k)
        assert ref.shape == (1, 1, 1, h, w, 2)
        out[f"{tag}.motion"], out[f"{tag}.depth"], out[f"{tag}.k"] = motion.numpy(), depth.numpy(), np.int32(k)
        out[f"{tag}.out"] = ref.reshape(1, h, w, 2).numpy()
    np.savez_compressed(os.path.join(OUT, "dilate_velocity.npz"), **out)


def gen_feeder():
    """Sample loading of the reference (core/data/supersampling_sequence.py:153-200): fp16 CHW `arr_0` arrays, crop,
    flips (with the motion-vector sign fix), cast to fp32, stacked over the clip -> tests/golden/feeder.npz.
    Runs the dataset class's own static methods on synthetic arrays written to a temporary directory."""
    import tempfile
    from core.data.supersampling_sequence import SuperSamplingDataset_Sequence as DS
    from core.type_hints import CropRect, TransformParams
    rng = np.random.default_rng(99)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        h, w = 22, 31
        frames = [rng.random((3, h, w), dtype=np.float32).astype(np.float16) for _ in range(2)]
        buffers = [rng.random((10, h, w), dtype=np.float32).astype(np.float16) for _ in range(2)]
        motion = [((rng.random((2, h, w), dtype=np.float32) - 0.5) * 8).astype(np.float16) for _ in range(2)]
        paths = {"frames": [], "buffers": [], "motion": []}
        for kind, arrs in (("frames", frames), ("buffers", buffers), ("motion", motion)):
            for i, a in enumerate(arrs):
                pth = os.path.join(tmp, f"{kind}_{i}.npz")
                np.savez_compressed(pth, a)
                paths[kind].append(pth)
                out[f"src.{kind}.{i}"] = a
        cases = {"plain": (None, (False, False)), "crop": ((5, 3, 5 + 16, 3 + 12), (False, False)),
                 "hflip": ((5, 3, 21, 15), (True, False)), "vflip": ((0, 0, 31, 22), (False, True)),
                 "both": ((7, 2, 7 + 9, 2 + 17), (True, True))}
        for name, (rect, (hf, vf)) in cases.items():
            crop = None if rect is None else CropRect(*rect)
            tr = None if rect is None else TransformParams(hf, vf, False)
            out[f"{name}.rect"] = np.array([-1, -1, -1, -1] if rect is None else rect, np.int64)
            out[f"{name}.flips"] = np.array([hf, vf], np.int64)
            for kind, fn in (("frames", DS._fn_transform), ("buffers", DS._fn_transform), ("motion", DS._fn_transform_velocity)):
                out[f"{name}.{kind}"] = DS._load_and_augment(paths[kind], crop, tr, fn).numpy()
    np.savez_compressed(os.path.join(OUT, "feeder.npz"), **out)
    print("feeder.npz:", len(out), "arrays")


def gen_temporal_model():
    """Full forward of the temporal variant, core/models/model_final.py SuperSampler(3, 10, 4, 1), at LR 16x24.
    The reference's backward_warping (core/tools/math.py:10-17) permutes an already (x, y)-last grid and
    raises for every input; it is replaced here by the evidently intended form (same arithmetic, no permute)."""
    torch.cuda.current_device = lambda: "cpu"
    import core.tools.math as mutil
    from torch.nn import functional as F

    def warp_fixed(frame, motion, mode="nearest", align_corners=True):
        h, w = frame.shape[-2:]
        gx = torch.arange(w).view(1, w).expand(h, -1).float()
        gy = torch.arange(h).view(h, 1).expand(-1, w).float()
        gx = ((gx[None, ...] + motion[:, 0]) / (w - 1)) * 2.0 - 1.0
        gy = ((gy[None, ...] + motion[:, 1]) / (h - 1)) * 2.0 - 1.0
        return F.grid_sample(frame, torch.stack([gx, gy], dim=-1), mode, "zeros", align_corners)

    mutil.backward_warping = warp_fixed
    import core.models.model_final as mf
    torch.manual_seed(42)
    m = mf.SuperSampler(3, 10, 4, 1)
    init = {name: np.concatenate([p.detach().reshape(-1)[:8].numpy(),
                                  np.array([p.detach().double().sum().item()], np.float32)]).astype(np.float32)
            for name, p in m.state_dict().items()}
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("bias"):
                p.copy_(bf16_round(torch.randn(p.shape, generator=g) * 0.05))
            elif p.dim() > 1:
                p.copy_(bf16_round(p))
    m.eval()
    b, f, h, w, s = 1, 1, 16, 24, 4
    lr_frame = rnd(g, b, 3, h, w, lo=0.0, hi=1.0)
    hr_buffer = rnd(g, b, 10, h * s, w * s, lo=0.0, hi=1.0)
    hr_buffer[:, 9] *= 5.0                       # depth beyond the first clamp threshold
    lr_buffer = F.interpolate(hr_buffer, scale_factor=1.0 / s, mode="nearest")
    motion = rnd(g, b, f, 2, h, w, lo=-2.0, hi=2.0)
    hist_frames = (lr_frame[:, None] + 0.05 * rnd(g, b, f, 3, h, w)).clamp(0, 1)
    hist_buffer = (lr_buffer[:, None] + 0.02 * rnd(g, b, f, 10, h, w)).clamp(0, 6)
    out = {"lr_frame": lr_frame.numpy(), "hr_buffer": hr_buffer.numpy(), "lr_buffer": lr_buffer.numpy(),
           "motion": motion.numpy(), "hist_frames": hist_frames.numpy(), "hist_buffer": hist_buffer.numpy()}
    with torch.no_grad():
        sr, ew = m(lr_frame.clone(), hr_buffer.clone(), lr_buffer.clone(), motion.clone(), hist_frames.clone(), hist_buffer.clone())
    out["sr"] = sr.numpy(); out["enhance_weight"] = ew.numpy()
    out["keys"] = np.array(list(m.state_dict().keys()))
    # 6.8 M weights are not stored: they are bf16_round(seed-42 initialisation), which the mirror reproduces
    # seed for seed (checked through the digests below); biases and the scalar parameter are stored.
    for k, v in m.state_dict().items():
        out["init." + k] = init[k]
        if v.dim() <= 1:
            out["small." + k] = v.detach().numpy()
        else:
            out["digest." + k] = np.array([v.detach().double().sum().item(), v.detach().double().abs().sum().item()])
    np.savez_compressed(os.path.join(OUT, "temporal_model.npz"), **out)
    print("temporal_model.npz: sr", tuple(sr.shape), "mean", sr.abs().mean().item(), "enhance_weight", tuple(ew.shape),
          "params", sum(p.numel() for p in m.parameters()))


def gen_temporal_train(h=48, w=64, fname="temporal_train.npz"):
    """Parameter gradients of the temporal variant under the reference's own autograd: the model and weights of
    temporal_model.npz (seed-42 initialisation rounded to bf16, its stored biases), seeded inputs at LR 48 x 64
    (temporal_train_inputs), loss = mean((1 + enhance_weight up-sampled) * |sr - target|) + mean(sr * G) with a fixed
    positive G, i.e. an L1 term weighted the way the variant's second output is meant to be used plus a dense
    well-conditioned term.  6.8 M gradients are not stored: per tensor its L2 norm, eight projections on fixed +-1
    vectors (seeded per tensor; the projections of an error estimate its norm) and, for tensors of at most 4,096
    elements, the tensor itself."""
    torch.cuda.current_device = lambda: "cpu"
    import core.tools.math as mutil
    from torch.nn import functional as F

    def warp_fixed(frame, motion, mode="nearest", align_corners=True):
        h, w = frame.shape[-2:]
        gx = torch.arange(w).view(1, w).expand(h, -1).float()
        gy = torch.arange(h).view(h, 1).expand(-1, w).float()
        gx = ((gx[None, ...] + motion[:, 0]) / (w - 1)) * 2.0 - 1.0
        gy = ((gy[None, ...] + motion[:, 1]) / (h - 1)) * 2.0 - 1.0
        return F.grid_sample(frame, torch.stack([gx, gy], dim=-1), mode, "zeros", align_corners)

    mutil.backward_warping = warp_fixed
    import core.models.model_final as mf
    z = np.load(os.path.join(OUT, "temporal_model.npz"))
    torch.manual_seed(42)
    m = mf.SuperSampler(3, 10, 4, 1)
    with torch.no_grad():
        for name, p in m.state_dict().items():
            if p.dim() > 1:
                p.copy_(bf16_round(p))
            else:
                p.copy_(torch.from_numpy(z["small." + name]))
    m.train()
    from seeded_inputs import temporal_train_inputs   # tests/golden/seeded_inputs.py: the GPU test builds the same tensors
    args, target, G = temporal_train_inputs(h, w)
    sr, ew = m(*[a.clone() for a in args])
    wmap = F.interpolate(ew, size=sr.shape[-2:], mode="nearest")
    loss = ((1.0 + wmap) * (sr - target).abs()).mean() + (sr * G).mean()
    loss.backward()
    out = {"loss": np.array([loss.item()], np.float32), "sr": sr.detach().numpy().astype(np.float16),
           "enhance_weight": ew.numpy()}
    names = []
    for i, (name, p) in enumerate(m.named_parameters()):
        gr = p.grad.detach().double().reshape(-1)
        names.append(name)
        gen = torch.Generator().manual_seed(1000 + i)
        proj = []
        for _ in range(8):
            v = (torch.randint(0, 2, (gr.numel(),), generator=gen).double() * 2.0 - 1.0)
            proj.append(float((gr * v).sum()))
        out["norm." + name] = np.array([float(gr.norm())])
        out["proj." + name] = np.array(proj)
        if gr.numel() <= 4096:
            out["grad." + name] = p.grad.detach().numpy()
    out["names"] = np.array(names)
    if fname != "temporal_train.npz":   # the small-size companion only serves the error-versus-pixel-count comparison
        out = {k: v for k, v in out.items() if k.startswith(("norm.", "proj.")) or k in ("names", "loss")}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", loss.item(), "tensors", len(names))


if __name__ == "__main__":
    _stub_modules()
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "reprfn_init":
        gen_reprfn_init()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "layernorm_bwd":
        gen_layernorm_bwd()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reprfn_train":
        gen_reprfn_train()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dilate_velocity":
        gen_dilate_velocity()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "feeder":
        gen_feeder()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "temporal_masks":
        gen_temporal_masks()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "temporal_model":
        gen_temporal_model()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "temporal_train":
        gen_temporal_train()
        gen_temporal_train(16, 24, "temporal_train_small.npz")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "temporal_ops":
        gen_temporal_ops()
        sys.exit(0)
    gen_ops()
    gen_layernorm_bwd()
    gen_supersampler()
    gen_reprfn()
    gen_reprfn_train()
    gen_reprfn_init()
    gen_temporal_ops()
    gen_temporal_masks()
    gen_feeder()
    gen_dilate_velocity()
    gen_temporal_model()
    gen_temporal_train()
    gen_temporal_train(16, 24, "temporal_train_small.npz")
