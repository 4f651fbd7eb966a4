"""Benchmark of the super-resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload infer4k|train]

Default workload (BASELINE.json configs[1]): 4x SR inference 540p -> 2160p (4K), batch 1, bf16
compute, default network (core/models/model.py SuperSampler), synthetic frames, seed-42 weights.
A step = one forward pass over one frame whose inputs are already resident in HBM.  The same run then
times the training step (the second half of the metric) and reports it in a `train` block of the line.  With N > 1
ranks (one process per GPU, launched by torch.distributed.run) frames are sharded across ranks
with no data-path collective (weak scaling); the value is whole-job frames/s.

`--workload train` (BASELINE.json configs[2]): one step = nearest down-sampling of the aliased crops +
forward + clip + L1 + 0.5*SSIM + backward + gradient all-reduce + AdamW on 16 HR crops of 1024x1024
per GPU (core/trainer.py:143-161 with sequence length 1); the value is whole-job crops/s.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the 64->64 3x3 MFMA
convolution at 4K: 21 of the 25 HR layers), timed per launch with events on the launch stream;
`cpu_baseline` times the CPU oracle on a bounded crop of the same workload on the host cores.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LR_H, LR_W, SCALE = 540, 960, 4
HR_H, HR_W = LR_H * SCALE, LR_W * SCALE
# algorithmic work of the default network per frame (conv FLOP = 2*MAC, bias/activation excluded)
FLOP_PER_HR_PX = 2 * 892224
FLOP_PER_LR_PX = 2 * 19296
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense
HBM_PEAK_GBS = 8000.0
# The dominant kernel (64->64 3x3, bf16 NHWC in and out) moves 2 x 128 B per pixel for 73,728 FLOP: 288 FLOP/B,
# under the machine balance of 2500 TFLOP/s / 8 TB/s = 312 FLOP/B, so HBM bandwidth is the roofline that bounds it.


def settle_host():
    """Called right before a timed region: collect garbage once and move every live object out of the collector's
    generations, so that no full collection (80-110 ms in a process that has imported torch) lands inside the timed
    steps.  The collector stays enabled; a training script does the same after building its model."""
    gc.collect()
    gc.freeze()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 40 / 100 for the millisecond-scale temporal / reprfn workloads)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 5; 40 / 100 for temporal / reprfn)")
    ap.add_argument("--workload", default="infer4k", choices=["infer4k", "train", "temporal", "temporal_train", "recurrent", "reprfn", "reprfn_train", "memops", "seq64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg of the default workload")
    ap.add_argument("--no-aux", action="store_true", help="skip the short runs of the other workloads in the default line")
    ap.add_argument("--train-steps", type=int, default=8, help="timed steps of the training leg of the default workload")
    ap.add_argument("--train-batch", type=int, default=16, help="crops per GPU per step (train workload)")
    ap.add_argument("--train-crop", type=int, default=1024, help="HR crop size (train workload)")
    ap.add_argument("--cpu-train-crop", type=int, default=1024, help="HR crop size of the CPU training-step sample")
    ap.add_argument("--seq-frames", type=int, default=64, help="frames of the sequence (seq64 workload), sharded over the ranks")
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[288, 384], help="LR crop (h w) of the CPU-oracle sample")
    args = ap.parse_args()
    # a step of these two networks takes 1-3 ms: five warm-up steps end before the GPU's clocks have settled, and the same
    # code then reads 5-8 % lower than it does inside the default line, where it runs behind seconds of other work
    steps, warmup = {"temporal": (40, 40), "reprfn": (100, 100)}.get(args.workload, (20, 5))
    args.steps = steps if args.steps is None else args.steps
    args.warmup = warmup if args.warmup is None else args.warmup
    return args


def make_model(dev):
    from superresolution_amd import SuperSampler
    torch.manual_seed(42)  # main.py:40 seeds with 42
    return SuperSampler(3, 10, 1).to(dev).eval()


def make_inputs(dev, rank):
    g = torch.Generator().manual_seed(42 + rank)
    lr = torch.rand(1, 2, 3, LR_H, LR_W, generator=g)
    hb = torch.rand(1, 2, 10, HR_H, HR_W, generator=g)
    hb[:, :, 9] *= 3.0  # depth channel range
    return lr.to(dev), hb.to(dev)


def dominant_kernel_time_ms(run_once, hr_h, frames):
    """Average duration of one launch of the plain 64->64 3x3 convolution at 4K as it runs inside
    the network (real activations), measured with events on the stream the kernel is launched on
    (torch's current stream): every such launch of `frames` extra forward passes is bracketed."""
    from superresolution_amd import functional as F_
    real = F_.conv3x3
    evs = []

    def timed(src_a, wpack, bias, cout, src_b=None, res=None, **kw):
        hit = src_b is None and res is None and src_a.shape[-1] == 64 and cout == 64 and src_a.shape[1] == hr_h
        if not hit:
            return real(src_a, wpack, bias, cout, src_b=src_b, res=res, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = real(src_a, wpack, bias, cout, src_b=src_b, res=res, **kw)
        b.record()
        evs.append((a, b))
        return out

    F_.conv3x3 = timed
    try:
        for _ in range(frames):
            run_once()
        torch.cuda.synchronize()
    finally:
        F_.conv3x3 = real
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2], len(ts)


def pair_kernel_time_ms(run_once, hr_h, frames):
    """Average duration of one launch of the fused residual-pair kernel (64 -> 64 -> 64, the dominant kernel of the
    inference pass: 9 launches = two thirds of a frame) as it runs inside the network, measured with events on the
    stream the kernel is launched on: every such launch of `frames` extra forward passes is bracketed."""
    from superresolution_amd import functional as F_
    real = F_.conv3x3_pair
    evs = []

    def timed(x, *a, **kw):
        if x.shape[1] != hr_h:
            return real(x, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real(x, *a, **kw)
        e1.record()
        evs.append((e0, e1))
        return out

    F_.conv3x3_pair = timed
    try:
        for _ in range(frames):
            run_once()
        torch.cuda.synchronize()
    finally:
        F_.conv3x3_pair = real
    if not evs:
        return None
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2], len(ts)


def pair_roofline_block(hr_h, hr_w, avg_ms, med_ms, launches):
    """Roofline of the fused pair kernel: 2 x 73,728 FLOP and 2 x 128 B (x in, dst out) per pixel = 576 FLOP/B, above
    the machine balance of 2500 TFLOP/s / 8 TB/s = 312 FLOP/B, so the bf16 MFMA peak is the roofline that bounds it."""
    px = float(hr_h * hr_w)
    flop, nbytes = 2.0 * px * 64 * 64 * 9 * 2, px * 256.0
    tflops = flop / (avg_ms * 1e-3) / 1e12
    traffic = measured_traffic_bytes("r3_bi_pair_pmc.csv", "conv3x3_pair_kernel<false, 0, 0, 0>")
    return {"bound": "mfma", "kernel": f"conv3x3_pair_kernel<false,*> (conv 64->64 + bias + ReLU -> conv 64->64 + bias [+ ReLU], x alpha, + x @ {hr_h}x{hr_w}; "
                                       "9 of the 16 HR launches per frame: 6 SR blocks <false,0> and 3 G-buffer residual pairs <false,1>, timed together)",
            "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / BF16_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes/launch (profiles/r3_bi_pair_pmc.csv)" if traffic is not None else None,
            "avg_launch_ms": avg_ms, "median_launch_ms": med_ms, "launches_timed": launches,
            "algorithmic_bytes_per_launch": nbytes, "algorithmic_flop_per_launch": flop,
            "hbm_view": {"achieved": nbytes / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def roofline_block(kernel, flop, nbytes, avg_ms, med_ms, launches, traffic, traffic_file="r1_g_conv64_pmc.csv"):
    """Roofline of the dominant kernel from its event-timed average launch duration.  Bound: HBM
    (arithmetic intensity 288 FLOP/B < 312 FLOP/B machine balance); the MFMA view is kept beside it."""
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    tflops = flop / (avg_ms * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": f"bytes/launch (profiles/{traffic_file})" if traffic is not None else None,
            "avg_launch_ms": avg_ms, "median_launch_ms": med_ms, "launches_timed": launches,
            "algorithmic_bytes_per_launch": nbytes, "algorithmic_flop_per_launch": flop,
            "mfma_view": {"achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": tflops / BF16_MFMA_PEAK_TFLOPS}}


def measured_traffic_bytes(name="r1_g_conv64_pmc.csv", kernel="Cfg<4, 2, 1, false, 0>"):
    """HBM bytes per launch of a kernel from the committed counter passes (separate
    FETCH_SIZE and WRITE_SIZE runs, KiB units; FETCH_SIZE doubled as gfx950 tallies the 128-byte
    requests of wide streaming reads at 64 bytes).  None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        vals = {}
        for line in open(path):
            parts = line.rstrip().rsplit(",", 3)
            if len(parts) == 4 and kernel in parts[0] and parts[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[parts[1]] = float(parts[3])
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def _cpu_threads():
    cores = min(16, os.cpu_count() or 1)   # the GPU box grants 16 host cores per GPU
    torch.set_num_threads(cores)
    return cores


def cpu_baseline(crop):
    """The reference's CPU path beside the GPU: the default network composed from the stock torch operators the
    reference itself calls (F.conv2d / F.grid_sample / torch.cat, fp32, ATen CPU kernels; oracle/torch_checkers.py,
    pinned to the reference's own output by the CPU suite) on ONE full 540x960 -> 2160x3840 frame, the very workload of
    `value` (BASELINE configs[1]); nothing is scaled.  A full 270x480 -> 1080x1920 frame (BASELINE configs[0], the
    reference's own CPU-runnable case) is timed next to it, and the C restatement (oracle/oracle.c) on a bounded crop."""
    import numpy as np
    from oracle import oracle as orc
    from oracle.torch_checkers import supersampler_forward_torch
    from superresolution_amd import SuperSampler
    cores = _cpu_threads()
    torch.manual_seed(42)
    sd_t = {k: v.detach().clone() for k, v in SuperSampler(3, 10, 1).state_dict().items()}
    g = torch.Generator().manual_seed(42)
    lr = torch.rand(1, 3, LR_H, LR_W, generator=g)
    hb = torch.rand(1, 10, HR_H, HR_W, generator=g)
    h1, w1 = LR_H // 2, LR_W // 2
    with torch.no_grad():
        supersampler_forward_torch(sd_t, lr[..., :16, :16], hb[..., :64, :64])  # page in
        t0 = time.perf_counter()
        supersampler_forward_torch(sd_t, lr, hb)
        dt = time.perf_counter() - t0
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            supersampler_forward_torch(sd_t, lr[..., :h1, :w1].contiguous(), hb[..., :h1 * SCALE, :w1 * SCALE].contiguous())
            ts.append(time.perf_counter() - t0)
    dt1 = min(ts)
    res = {"value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"stock torch CPU operators (F.conv2d / F.grid_sample / torch.cat, fp32, {cores} threads: the calls the "
                     f"reference's nn.Modules make) composing core/models/model.py:147-175 on one full {LR_H}x{LR_W} -> {HR_H}x{HR_W} "
                     f"frame, one timed pass ({dt:.2f} s), unscaled",
           "config0": {"value": 1.0 / dt1, "unit": "frames/s of 270x480 -> 1080x1920 (BASELINE configs[0])", "cores": cores,
                       "sample": f"the same composition on one full {h1}x{w1} -> {h1 * SCALE}x{w1 * SCALE} frame, best of 2 ({dt1:.2f} s)"}}
    # the C restatement on its bounded crop
    orc.set_threads(cores)
    sd = {k: v.numpy() for k, v in sd_t.items()}
    ch, cw = crop
    rng = np.random.default_rng(42)
    lr_c = rng.random((1, 1, 3, ch, cw), dtype=np.float32)
    hb_c = rng.random((1, 1, 10, ch * SCALE, cw * SCALE), dtype=np.float32)
    orc.supersampler_forward(sd, lr_c[:, :, :, :8, :8], hb_c[:, :, :, :32, :32])  # page in
    t0 = time.perf_counter()
    orc.supersampler_forward(sd, lr_c, hb_c)
    dto = time.perf_counter() - t0
    res["oracle"] = {"value": (ch * cw) / float(LR_H * LR_W) / dto, "unit": "frames/s", "cores": cores, "kind": "port",
                     "sample": f"oracle/oracle.c forward on one LR {ch}x{cw} -> HR {ch * SCALE}x{cw * SCALE} crop ({dto:.2f} s), "
                               f"scaled by pixel count to a 540x960 -> 2160x3840 frame"}
    res["ops"] = cpu_baseline_ops(cores)
    return res


def cpu_baseline_ops(cores):
    """The two memory-bound neighbours the GPU line carries as `warp_gbps` / `pixel_shuffle_gbps`, with the stock torch CPU
    operators the reference calls (core/tools/math.py:10-17 -> F.grid_sample; core/arch/RepRFN.py:272 -> PixelShuffle) at
    the same 4K shapes: algorithmic GB/s, median of 3."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    fr = torch.rand(1, 3, HR_H, HR_W, generator=g)
    mv = (torch.randn(1, 2, HR_H, HR_W, generator=g) * 4).clamp(-32, 32)

    def warp():
        col = torch.arange(HR_W, dtype=torch.float32).view(1, 1, HR_W)
        row = torch.arange(HR_H, dtype=torch.float32).view(1, HR_H, 1)
        gx = ((col + mv[:, 0]) / (HR_W - 1)) * 2 - 1
        gy = ((row + mv[:, 1]) / (HR_H - 1)) * 2 - 1
        return F.grid_sample(fr, torch.stack((gx, gy), dim=-1), mode="bilinear", padding_mode="zeros", align_corners=True)

    x = torch.rand(1, 48, LR_H, LR_W, generator=g)

    def med(fn):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]

    with torch.no_grad():
        tw, tp = med(warp), med(lambda: F.pixel_shuffle(x, 4))
    return {"warp_gbps": (2 * 3 + 2) * HR_H * HR_W * 4 / tw / 1e9, "pixel_shuffle_gbps": 2 * 48 * LR_H * LR_W * 4 / tp / 1e9,
            "cores": cores, "sample": "F.grid_sample warp of a 3-channel fp32 2160x3840 frame by per-pixel vectors (grid built as "
                                      "core/tools/math.py:12-15) and F.pixel_shuffle of (1,48,540,960) fp32, median of 3"}


def cpu_baseline_train(crop):
    """One whole training step (core/trainer.py:143-161: forward, clip, L1 + 0.5 SSIM, backward, AdamW) on ONE HR crop
    (batch 1: a batch of 16 needs > 100 GB of fp32 activations) with stock torch operators and torch autograd on the
    host cores, median of 3, scaled by pixel count to 1024x1024 crops/s."""
    from oracle.torch_checkers import LossGroup, supersampler_forward_torch
    from superresolution_amd import SuperSampler
    cores = _cpu_threads()
    torch.manual_seed(42)
    model = SuperSampler(3, 10, 1)
    sd = dict(model.named_parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    lg = LossGroup(1.0, 0.5, report=False)
    g = torch.Generator().manual_seed(42)
    alias = torch.rand(1, 3, crop, crop, generator=g)
    target = torch.rand(1, 3, crop, crop, generator=g)
    hb = torch.rand(1, 10, crop, crop, generator=g)

    def step(n):
        lr = torch.nn.functional.interpolate(alias[..., :n, :n], scale_factor=1.0 / SCALE, mode="nearest")
        sr = torch.clip(supersampler_forward_torch(sd, lr, hb[..., :n, :n]), 0.0, 10.0)
        opt.zero_grad()
        loss, _ = lg({"hr_frames": target[..., :n, :n], "sr_frames": sr})
        loss.backward()
        opt.step()

    step(64)  # page in
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        step(crop)
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    return {"value": (crop * crop) / (1024.0 * 1024.0) / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"stock torch CPU operators + torch autograd (fp32, {cores} threads): nearest down-sampling + forward + clip + "
                      f"L1 + 0.5*SSIM + backward + AdamW on one HR {crop}x{crop} crop, median of 3 ({dt:.2f} s), scaled by pixel "
                      f"count to 1024x1024 crops"}


def run_train(args, world, rank, dev, dist, steps=None, warmup=None, with_cpu=True):
    """Times the training step on every rank; returns the result line (rank 0) or None."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from superresolution_amd import functional as F_
    from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
    from superresolution_amd.loss import FusedLossGroup
    B, S = args.train_batch, args.train_crop
    model = make_model(dev).train()
    broadcast_parameters(model, src=0)
    # two buckets, each all-reduced asynchronously as soon as the backward pass has queued its last weight gradient
    red = FlatGradAllReducer(model.parameters(), buckets=2).attach(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)  # configs/trainer/4x_H2.yaml:12-17
    loss_group = FusedLossGroup(1.0, 0.5, clip=(0.0, 10.0), report=False)  # clip: core/trainer.py:154-156
    g = torch.Generator().manual_seed(42 + rank)
    alias = torch.rand(B, 1, 3, S, S, generator=g).to(dev)
    target = torch.rand(B, 1, 3, S, S, generator=g).to(dev)
    buffers = torch.rand(B, 1, 10, S, S, generator=g)
    buffers[:, :, 9] *= 3.0
    buffers = buffers.to(dev)

    def step():
        # core/trainer.py:86-113 (_make_inputs, scale 4) and :143-161 (one frame per clip)
        lr = F_.nearest_down(alias.view(B, 3, S, S), 1.0 / SCALE).view(B, 1, 3, S // SCALE, S // SCALE)
        batch = {"scale": float(SCALE), "hr_frames": target[:, 0], "lr_frames": lr, "hr_buffers": buffers}
        batch = model(batch)
        red.zero_grad()
        loss, _ = loss_group(batch)
        loss.backward()
        red.all_reduce()
        opt.step()
        return loss

    loss0 = float(step().detach())
    for _ in range(max(warmup - 1, 0)):
        step()
    settle_host()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # a step timed on NaN tensors runs at different clocks: the loss must be finite and must have fallen on
    # this fixed batch, and every gradient of the last step must be finite
    assert bool(torch.isfinite(loss)) and float(loss.detach()) < loss0, f"training diverged: {loss0} -> {float(loss.detach())}"
    assert bool(torch.isfinite(red.flat).all()), "non-finite gradients"
    # every rank runs the instrumented extra steps: they contain the gradient all-reduce
    k_avg_ms, k_med_ms, k_launches = dominant_kernel_time_ms(step, S, 2)
    if rank != 0:
        return None
    ms_per_step = dt / steps * 1e3
    p_hr, p_lr = B * S * S, B * (S // SCALE) ** 2
    fwd = FLOP_PER_HR_PX * p_hr + FLOP_PER_LR_PX * p_lr
    # backward = dgrad + wgrad of every layer except the data gradient of the two input layers
    no_dgrad = 2 * 9 * (10 * 64 * p_hr + 3 * 32 * p_lr)
    flop_step = 3 * fwd - no_dgrad
    k_flop = 2.0 * B * S * S * 64 * 64 * 9
    res = {
        "metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs",
        "value": world * B * steps / dt, "unit": "crops/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"training step on {B} HR crops of {S}x{S} per GPU (LR {S // SCALE}x{S // SCALE}, scale 4): "
                               "nearest down-sampling + forward + clip + L1 + 0.5*SSIM + backward + gradient all-reduce + AdamW, "
                               "default SuperSampler, fp32 master weights",
                   "crops_per_step_per_gpu": B, "tflop_per_step_per_gpu": flop_step / 1e12,
                   "net_tflops_per_gpu": flop_step / 1e12 / (ms_per_step * 1e-3),
                   "net_frac_of_bf16_mfma_peak": flop_step / 1e12 / (ms_per_step * 1e-3) / BF16_MFMA_PEAK_TFLOPS},
        "roofline": roofline_block(f"conv3x3_kernel<Cfg<4,2,1,false,0>> (64->64 3x3 + bias + ReLU @ {B}x{S}x{S}; the same kernel "
                                   "runs every data gradient)", k_flop, 2.0 * B * S * S * 64 * 2, k_avg_ms, k_med_ms,
                                   k_launches,
                                   measured_traffic_bytes("r2_m_train_kernels_pmc.csv", "conv3x3_kernel<") if (B, S) == (16, 1024) else None,
                                   "r2_m_train_kernels_pmc.csv"),
    }
    if with_cpu and not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_train(args.cpu_train_crop)
    return res


def run_reprfn_train(args, world, rank, dev, dist):
    """Training step of RepRFN in its training form (nine branches per block, configs/trainer/4x_RepRFN.yaml) on
    `train_batch` LR crops of (train_crop/4)^2 per GPU: forward + L1 + backward + gradient all-reduce + AdamW."""
    from superresolution_amd import RepRFN
    from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
    B, S = args.train_batch, args.train_crop // SCALE
    torch.manual_seed(42)
    model = RepRFN().to(dev).train()
    broadcast_parameters(model, src=0)
    params = [p for p in model.parameters() if p.requires_grad]
    # 264 small parameters: one multi-tensor copy per bucket instead of 264 adds; the two buckets start from inside backward
    red = FlatGradAllReducer(params, gather=True, buckets=2).attach_hooks()
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(42 + rank)
    x = torch.rand(B, 3, S, S, generator=g).to(dev)
    target = torch.rand(B, 3, S * SCALE, S * SCALE, generator=g).to(dev)

    def step():
        red.zero_grad()
        loss = (model(x) - target).abs().mean()
        loss.backward()
        red.all_reduce()
        opt.step()
        return loss.detach()

    loss0 = float(step())
    for _ in range(max(args.warmup - 1, 0)):
        step()
    settle_host()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss)) and float(loss) < loss0 and bool(torch.isfinite(red.flat).all())
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    # deploy-form convolution FLOPs per LR pixel (0.404 TFLOP per 540x960 frame), x3 for forward + both gradients
    flop_step = 3.0 * 0.40421293056e12 / (LR_H * LR_W) * B * S * S
    tflops = flop_step / 1e12 / (ms * 1e-3)
    return {"metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs", "value": world * B * args.steps / dt,
            "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"RepRFN(3,3,48,4) training form, {B} LR crops of {S}x{S} (HR {S * SCALE}x{S * SCALE}) per GPU: forward + L1 + "
                                   "backward + gradient all-reduce + AdamW, fp32 master weights", "crops_per_step_per_gpu": B,
                       "tflop_per_step_per_gpu": flop_step / 1e12},
            "roofline": {"bound": "mfma", "kernel": "whole step (merged-kernel convolution FLOPs, forward + data + weight gradients, "
                         "over the wall time of a step)", "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / BF16_MFMA_PEAK_TFLOPS, "traffic": None}}


def run_temporal_train(args, world, rank, dev, dist):
    """Training step of the temporal SuperSampler (core/models/model_final.py, n_history 1) on `train_batch` clips of one
    history frame per GPU, LR (train_crop/4)^2: forward + enhance-weighted L1 (the variant's second output is a per-pixel
    loss weight, model_final.py:499-505) + backward + gradient all-reduce + AdamW, data-parallel like the other two
    networks (one flat gradient buffer, one collective)."""
    from superresolution_amd.dp import FlatGradAllReducer, broadcast_parameters
    from superresolution_amd.temporal import SuperSampler as TemporalSuperSampler
    from superresolution_amd.loss import enhance_weighted_l1
    B, S = args.train_batch, args.train_crop // SCALE
    torch.manual_seed(42)
    model = TemporalSuperSampler(3, 10, SCALE, 1).to(dev).train()
    broadcast_parameters(model, src=0)
    params = [p for p in model.parameters() if p.requires_grad]
    # 27 MB of gradients in four buckets, each reduced as soon as autograd has finished its last parameter (reverse layer order)
    red = FlatGradAllReducer(params, gather=True, buckets=4).attach_hooks()
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(42 + rank)
    ins = [torch.rand(B, 3, S, S, generator=g), torch.rand(B, 10, S * SCALE, S * SCALE, generator=g), torch.rand(B, 10, S, S, generator=g),
           torch.randn(B, 1, 2, S, S, generator=g), torch.rand(B, 1, 3, S, S, generator=g), torch.rand(B, 1, 10, S, S, generator=g)]
    ins = [t.to(dev) for t in ins]
    target = torch.rand(B, 3, S * SCALE, S * SCALE, generator=g).to(dev)
    # the depth compression rewrites the depth planes of the three buffer inputs in place and is not idempotent
    depth = [ins[1][:, -1].clone(), ins[2][:, -1].clone(), ins[5][:, :, -1].clone()]

    def step():
        ins[1][:, -1].copy_(depth[0]); ins[2][:, -1].copy_(depth[1]); ins[5][:, :, -1].copy_(depth[2])
        red.zero_grad()
        sr, ew = model(*ins)
        # ((1 + nearest_upsample(ew)) * (sr - target).abs()).mean(), value and gradient in one HIP call
        loss = enhance_weighted_l1(sr, target, ew)
        loss.backward()
        red.all_reduce()
        opt.step()
        return loss.detach()

    with _FlopCounter() as fc, torch.no_grad():
        model.infer(*[t.clone() for t in ins])
    loss0 = float(step())
    for _ in range(max(args.warmup - 1, 0)):
        step()
    settle_host()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss)) and float(loss) < loss0 and bool(torch.isfinite(red.flat).all())
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    flop_step = 3.0 * fc.flop        # forward + data gradient + weight gradient of the forward's convolution FLOPs
    tflops = flop_step / 1e12 / (ms * 1e-3)
    return {"metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs", "value": world * B * args.steps / dt,
            "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"temporal SuperSampler(3,10,4,1) training, {B} clips of LR {S}x{S} (HR {S * SCALE}x{S * SCALE}) with one history "
                                   "frame per GPU: forward + enhance-weighted L1 + backward + gradient all-reduce + AdamW, fp32 master weights",
                       "crops_per_step_per_gpu": B, "tflop_per_step_per_gpu": flop_step / 1e12},
            "roofline": {"bound": "mfma", "kernel": "whole step (the forward's convolution FLOPs x 3 over the wall time of a step)",
                         "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / BF16_MFMA_PEAK_TFLOPS,
                         "traffic": None}}


class _FlopCounter:
    """Algorithmic convolution FLOPs of one forward pass, counted at the operator boundary (2 * Cout * Cin * kh * kw
    per output pixel with the layer's REAL kernel size and stride; the extra work of composed launches is not
    counted)."""

    def __init__(self):
        from superresolution_amd import functional as F_
        self.F, self.flop, self.depth, self.saved = F_, 0.0, 0, {}

    def __enter__(self):
        F_ = self.F

        def wrap(name, count):
            fn = getattr(F_, name)
            self.saved[name] = fn

            def run(*a, **k):
                self.depth += 1
                try:
                    out = fn(*a, **k)
                finally:
                    self.depth -= 1
                if self.depth == 0:
                    self.flop += count(out, *a, **k)
                return out
            setattr(F_, name, run)

        def px(out, nchw=False):
            return out.shape[0] * (out.shape[2] * out.shape[3] if nchw else out.shape[1] * out.shape[2])
        wrap("conv2d_auto", lambda out, x, w, *a, **k: 2.0 * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3] * px(out))
        wrap("conv2d_direct", lambda out, x, w, *a, **k: 2.0 * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3] * px(out))
        wrap("conv_transpose2x2s2", lambda out, x, w, *a, **k: 2.0 * w.shape[0] * w.shape[1] * px(out))
        wrap("conv3x3", lambda out, x, wp, b, cout, *a, **k: 18.0 * (x.shape[-1] + (k["src_b"].shape[-1] if k.get("src_b") is not None else 0))
             * cout * px(out, k.get("out_nchw_f32", False)))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.F, name, fn)


def run_aux_infer(args, world, rank, dev, dist):
    """Frames/s of the other two networks of the path at the 4K output size: RepRFN (core/arch/RepRFN.py, deploy
    form, 540x960 -> 2160x3840) and the temporal SuperSampler (core/models/model_final.py, n_history 1,
    544x960 -> 2176x3840: its encoder needs sizes divisible by 16).  Frames are sharded by rank, no collective."""
    g = torch.Generator().manual_seed(42 + rank)
    torch.manual_seed(42)
    reset = None   # recurrent workloads: puts the clip back to its first frame
    if args.workload == "recurrent":
        # BASELINE configs[3], first form: the previous 4K OUTPUT warped by HR motion vectors (bilinear reprojection) and
        # concatenated in front of the conv trunk.  One step = one frame of a clip: fused warp + concat + layout pass,
        # the default network on 13 buffer channels, and the clip to [0, 10] the trainer applies to the frame it feeds
        # back (core/trainer.py:154-156,163).  Motion ~ N(0, 4 px) clipped to +-32 px, 1 % of the vectors out of frame.
        from superresolution_amd import RecurrentSuperSampler
        model = RecurrentSuperSampler(3, 10, 1).to(dev).eval()
        lr = torch.rand(1, 2, 3, LR_H, LR_W, generator=g).to(dev)
        hb = torch.rand(1, 2, 10, HR_H, HR_W, generator=g)
        hb[:, :, 9] *= 3.0
        hb = hb.to(dev)
        mv = (torch.randn(1, 1, 2, HR_H, HR_W, generator=g) * 4.0).clamp_(-32.0, 32.0)
        far = torch.rand(1, 1, 1, HR_H, HR_W, generator=g) < 0.01
        mv = torch.where(far, torch.full_like(mv, 5000.0), mv).to(dev)
        clip_state = {"prev": None}

        def step():
            batch = {"lr_frames": lr, "hr_buffers": hb, "velocity": mv}
            if clip_state["prev"] is not None:
                batch["sr_frames"] = clip_state["prev"]
            out = model(batch)["sr_frames"]
            clip_state["prev"] = torch.clip(out, min=0.0, max=10.0)
            return out

        def reset():
            clip_state["prev"] = None
        out_shape = (1, 3, HR_H, HR_W)
        what = ("recurrent default SuperSampler: previous 4K output warped by HR motion vectors (bilinear, align_corners) and "
                "concatenated before the trunk, 540x960 -> 2160x3840, batch 1 per GPU, clips sharded by rank")
    elif args.workload == "reprfn":
        from superresolution_amd import RepRFN
        model = RepRFN().to(dev).eval()
        for mod in model.modules():
            if mod is not model and hasattr(mod, "switch_to_deploy"):
                mod.switch_to_deploy()
        x = torch.rand(1, 3, LR_H, LR_W, generator=g).to(dev)
        step = lambda: model.infer(x)
        out_shape = (1, 3, HR_H, HR_W)
        what = "RepRFN(3,3,48,4) deploy form, 540x960 -> 2160x3840, batch 1 per GPU, frames sharded by rank"
    else:
        from superresolution_amd.temporal import SuperSampler as TemporalSuperSampler
        model = TemporalSuperSampler(3, 10, 4, 1).to(dev).eval()
        h, w = 544, 960
        lr = torch.rand(1, 3, h, w, generator=g).to(dev)
        hb = torch.rand(1, 10, 4 * h, 4 * w, generator=g).to(dev)
        lb = torch.rand(1, 10, h, w, generator=g).to(dev)
        mv = torch.randn(1, 1, 2, h, w, generator=g).to(dev)
        hf = torch.rand(1, 1, 3, h, w, generator=g).to(dev)
        hbuf = torch.rand(1, 1, 10, h, w, generator=g).to(dev)
        # the depth compression rewrites the depth channel of its three buffer inputs in place (model_final.py:462) and is
        # not idempotent: every step starts from the original depth planes (restoring only those, not whole-tensor clones)
        d_hb, d_lb, d_hbuf = hb[:, -1].clone(), lb[:, -1].clone(), hbuf[:, :, -1].clone()

        def step():
            hb[:, -1].copy_(d_hb)
            lb[:, -1].copy_(d_lb)
            hbuf[:, :, -1].copy_(d_hbuf)
            return model.infer(lr, hb, lb, mv, hf, hbuf)[0]
        out_shape = (1, 3, 4 * h, 4 * w)
        what = ("temporal SuperSampler(3,10,4,1) (model_final.py), 544x960 -> 2176x3840 with one warped history frame, "
                "batch 1 per GPU, clips sharded by rank")
    with torch.no_grad():
        with _FlopCounter() as fc:
            first = step().clone()
        for _ in range(args.warmup):
            step()
        settle_host()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert tuple(out.shape) == out_shape and bool(torch.isfinite(out).all())
    if reset is None:
        assert torch.equal(out, first)
    else:   # the same clip walked again from its first frame ends on the same bits
        with torch.no_grad():
            reset()
            for _ in range(1 + args.warmup + args.steps):
                again = step()
        torch.cuda.synchronize()
        assert torch.equal(out, again)
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    if args.workload == "recurrent":
        # the default network's per-pixel work (core/models/model.py:98-115) plus three more input channels of its
        # first HR layer; the operator-boundary counter does not see inside the fused pair launches
        fc.flop = float((FLOP_PER_HR_PX + 2 * 64 * 3 * 9) * HR_H * HR_W + FLOP_PER_LR_PX * LR_H * LR_W)
    tflops = fc.flop / 1e12 / (ms * 1e-3)
    return {"metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs", "value": world * args.steps / dt,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": what, "frames_per_step_per_gpu": 1, "tflop_per_frame": fc.flop / 1e12},
            "roofline": {"bound": "mfma", "kernel": "whole network (many small launches; the convolutions' algorithmic FLOPs over "
                         "the wall time of a frame)", "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / BF16_MFMA_PEAK_TFLOPS, "traffic": None}}


def run_seq64(args, world, rank, dev, dist):
    """BASELINE configs[4]: batched offline up-scaling of a synthetic 64-frame sequence, frame i -> GPU i mod world
    (superresolution_amd.dp.shard_range), no collective.  The frames are DISTINCT and live in pinned host memory as fp16
    CHW arrays (the reference's on-disk format, core/data/supersampling_sequence.py:153-169; batch layout
    core/trainer.py:86-113: the network reads the last frame of `lr_frames` / `hr_buffers`).  Two timings of the whole
    shard, max over ranks: `value` with the shard already resident in HBM (decode + network per frame), and
    `with_upload` where every frame crosses PCIe on a side stream, double buffered, while the previous one computes."""
    from superresolution_amd.dp import shard_range
    from superresolution_amd.feeder import FrameStreamer, decode_crop_flip
    model = make_model(dev)
    mine = list(shard_range(args.seq_frames, rank, world))
    frames = []
    for i in mine:   # generated on the GPU (seeded per frame index, whatever rank owns it), parked in pinned host memory
        g = torch.Generator(device=dev).manual_seed(1000 + i)
        lr16 = torch.rand(3, LR_H, LR_W, device=dev, generator=g).half()
        hb16 = torch.rand(10, HR_H, HR_W, device=dev, generator=g).half()
        frames.append({"lr": lr16.cpu().pin_memory(), "hb": hb16.cpu().pin_memory()})
    del lr16, hb16
    torch.cuda.empty_cache()

    def sums_of(outs):
        return torch.stack(outs).cpu().tolist()

    # (a) shard resident in HBM as fp16
    resident = [{k: t.to(dev) for k, t in fr.items()} for fr in frames]
    lr32 = torch.empty(1, 3, LR_H, LR_W, device=dev)
    hb32 = torch.empty(1, 10, HR_H, HR_W, device=dev)

    def pass_resident():
        outs = []
        for fr in resident:
            decode_crop_flip(fr["lr"], out=lr32[0])
            decode_crop_flip(fr["hb"], out=hb32[0])
            outs.append(model.infer(lr32, hb32).double().sum())
        return outs

    streamer = FrameStreamer(frames, dev)

    def pass_streamed():
        return [model.infer(fr["lr"], fr["hb"]).double().sum() for fr in streamer]

    def timed(fn):
        fn()   # warm-up pass over the shard
        settle_host()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, sums_of(outs)

    with torch.no_grad():
        dt_res, sums_res = timed(pass_resident)
        dt_up, sums_up = timed(pass_streamed)
    assert all(map(lambda v: v == v and abs(v) < 1e12, sums_res)) and sums_res == sums_up, "streamed frames differ from resident ones"
    assert len(set(sums_res)) == len(sums_res), "the frames of the sequence are not distinct"
    if rank != 0:
        return None
    n = args.seq_frames
    flop_frame = FLOP_PER_HR_PX * HR_H * HR_W + FLOP_PER_LR_PX * LR_H * LR_W
    tflops = n * flop_frame / 1e12 / dt_res / world
    host_bytes = sum(t.numel() * 2 for t in frames[0].values())
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_memops
    res = {"metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs", "value": n / dt_res, "unit": "frames/s",
           "n_gpus": world, "steps": n, "warmup": len(mine), "ms_per_step": dt_res / len(mine) * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"offline up-scaling of a {n}-frame sequence of distinct 540x960 -> 2160x3840 frames, frame i on GPU i mod "
                                  f"{world}, shard resident in HBM as fp16 CHW (decode + default SuperSampler per frame)",
                      "frames": n, "frames_per_gpu": len(mine), "tflop_per_frame": flop_frame / 1e12},
           "with_upload": {"value": n / dt_up, "unit": "frames/s", "ms_per_frame_per_gpu": dt_up / len(mine) * 1e3,
                           "host_bytes_per_frame": host_bytes, "pcie_gbps_per_gpu": host_bytes * len(mine) / dt_up / 1e9,
                           "sample": "the same shard streamed from pinned host memory: frame k+1 uploads on a side stream into the other "
                                     "of two staging sets while frame k computes (superresolution_amd.feeder.FrameStreamer)"},
           "roofline": {"bound": "mfma", "kernel": "whole network per GPU (default SuperSampler, 14.82 TFLOP per frame, over the wall time "
                        "of the shard incl. the fp16 decode)", "achieved": tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / BF16_MFMA_PEAK_TFLOPS, "traffic": None}}
    res.update(bench_memops.headline())
    return res


def run_other_workloads(args, dev):
    """Short runs of the other workloads of the path (each is `--workload NAME` on its own) so that one default line
    carries a number and a roofline fraction for every network: same code, same validation (finite, bit-reproducible /
    loss falling), fewer timed steps.  Single GPU only; a failure is recorded, the headline is printed regardless."""
    import argparse
    out = {}
    for name, steps, warmup in (("temporal", 20, 5), ("reprfn", 40, 10), ("recurrent", 12, 3), ("temporal_train", 6, 2), ("reprfn_train", 12, 3),
                                ("seq64", 64, 64)):
        a = argparse.Namespace(**vars(args))
        a.workload, a.steps, a.warmup = name, steps, warmup
        try:
            if name == "temporal_train":
                r = run_temporal_train(a, 1, 0, dev, None)
            elif name == "reprfn_train":
                r = run_reprfn_train(a, 1, 0, dev, None)
            elif name == "seq64":
                r = run_seq64(a, 1, 0, dev, None)
            else:
                r = run_aux_infer(a, 1, 0, dev, None)
            out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": steps,
                         "frac_of_bf16_mfma_peak": r["roofline"]["frac"], "workload": r["config"]["workload"]}
            if "with_upload" in r:
                out[name]["with_upload"] = {k: r["with_upload"][k] for k in ("value", "unit", "pcie_gbps_per_gpu")}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def emit(args, res):
    """Rank 0's ONE JSON line; carries the number of ranks that really ran and the collective backend."""
    res.update(args.ranks_info)
    print(json.dumps(res), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks (one process per GPU) through
    torch.distributed.run BEFORE this process touches a GPU, relay rank 0's JSON line and return the launcher's exit
    code.  Counting devices does not initialise the GPU runtime."""
    import socket
    import subprocess
    backend = os.environ.get("SRHIP_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stdout.write(proc.stdout)
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {proc.returncode})", file=sys.stderr, flush=True)
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if os.environ.get("SRHIP_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SRHIP_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
    backend = os.environ.get("SRHIP_DIST_BACKEND", "nccl") if world > 1 else None
    if backend == "nccl" and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr, flush=True)
        sys.exit(2)
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") with one GPU per rank; SRHIP_DIST_BACKEND=gloo rehearses the multi-rank path on a
        # box with fewer GPUs than ranks (ranks then share a device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    args.ranks_info = {"ranks": dist.get_world_size() if dist is not None else 1, "backend": backend}

    if args.workload == "train":
        res = run_train(args, world, rank, dev, dist)
        if rank == 0:
            emit(args, res)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload == "memops":
        # the HBM-bound operators of the path at the 4K shapes, one launch each (tools/bench_memops.py): warp, pixel
        # (un)shuffle, up-sampling, layout, concat, nearest down-sampling, ReLU mask, fused loss
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_memops
        ops = bench_memops.suite(verbose=False)
        warp = next(o for o in ops if o["op"].startswith("warp bilinear f32"))
        if rank == 0:
            emit(args, {"metric": "algorithmic GB/s of the HBM-bound operators (warp of the previous 4K output first)",
                              "value": warp["gbps"], "unit": "GB/s", "n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": warp["ms"],
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "backward warp of a 3-channel fp32 2160x3840 frame by per-pixel motion vectors "
                                                     "(bilinear, align_corners=True) and the other memory-bound operators at the 4K shapes"},
                              "roofline": {"bound": "hbm", "kernel": "sample_f32_kernel<1> (backward warp)", "achieved": warp["gbps"],
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": warp["gbps"] / HBM_PEAK_GBS, "traffic": None},
                              "ops": ops})
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload == "temporal_train":
        res = run_temporal_train(args, world, rank, dev, dist)
        if rank == 0:
            emit(args, res)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload == "reprfn_train":
        res = run_reprfn_train(args, world, rank, dev, dist)
        if rank == 0:
            emit(args, res)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload == "seq64":
        res = run_seq64(args, world, rank, dev, dist)
        if rank == 0:
            emit(args, res)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload in ("temporal", "recurrent", "reprfn"):
        res = run_aux_infer(args, world, rank, dev, dist)
        if rank == 0:
            emit(args, res)
        if dist is not None:
            dist.destroy_process_group()
        return

    model = make_model(dev)
    lr, hb = make_inputs(dev, rank)

    def step():
        return model.infer(lr[:, -1], hb[:, -1])

    first = step().clone()  # the timed frames must reproduce it bit for bit (same inputs)
    for _ in range(args.warmup):
        step()
    settle_host()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # a number measured on garbage is worthless: finite, of sane magnitude, and identical to the first pass
    assert out.shape == (1, 3, HR_H, HR_W) and bool(torch.isfinite(out).all())
    assert float(out.abs().max()) < 1e3 and torch.equal(out, first), "inference output is not reproducible"
    del first

    # every rank takes part in the instrumented passes and in the training leg (gradient all-reduce)
    pair_t = pair_kernel_time_ms(step, HR_H, 5) if model.use_pair_kernel else None
    k_avg_ms, k_med_ms, k_launches = dominant_kernel_time_ms(step, HR_H, 5)
    train = None
    if not args.no_train:
        del out
        torch.cuda.empty_cache()
        try:
            train = run_train(args, world, rank, dev, dist, steps=args.train_steps, warmup=2, with_cpu=True)
            if train is not None:
                train = {k: train[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "config", "roofline", "cpu_baseline") if k in train}
        except Exception as e:  # the headline line must still be printed
            if world > 1:
                raise       # a rank that dropped out of a collective cannot be papered over
            train = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        fps = world * args.steps / dt
        flop_frame = FLOP_PER_HR_PX * HR_H * HR_W + FLOP_PER_LR_PX * LR_H * LR_W
        k_flop = 2.0 * HR_H * HR_W * 64 * 64 * 9
        res = {
            "metric": "4K output frames/sec (4x fwd, 1 GPU) + train crops/sec at 1/2/4/8 GPUs",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "4x SR inference 540x960 -> 2160x3840, default SuperSampler (4x_H2), batch 1 per GPU, "
                                   "frames sharded by rank", "frames_per_step_per_gpu": 1,
                       "tflop_per_frame": flop_frame / 1e12,
                       "net_tflops": flop_frame / 1e12 / (ms_per_step * 1e-3),
                       "net_frac_of_bf16_mfma_peak": flop_frame / 1e12 / (ms_per_step * 1e-3) / BF16_MFMA_PEAK_TFLOPS},
        }
        single = roofline_block("conv3x3_kernel<Cfg<4,2,1,false,0>> (64->64 3x3 + bias + ReLU @ 2160x3840, "
                                "3 of the 16 HR launches per frame)", k_flop, 2.0 * HR_H * HR_W * 64 * 2,
                                k_avg_ms, k_med_ms, k_launches, measured_traffic_bytes())
        if pair_t is not None:
            res["roofline"] = pair_roofline_block(HR_H, HR_W, *pair_t)
            res["roofline_single_layer"] = single
        else:
            res["roofline"] = single
        if train is not None:
            res["train"] = train   # second half of the metric: train crops/s at this GPU count (weak scaling)
        # the memory-bound neighbours of the path at the 4K shapes (algorithmic GB/s; `--workload memops` has them all)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_memops
        res.update(bench_memops.headline())
        if world == 1 and not args.no_aux:
            res["other_workloads"] = run_other_workloads(args, dev)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.cpu_crop)
        if world > 1:
            res["note"] = ("rank 0's line; value = frames of all ranks / max-over-ranks time; cpu_baseline and other_workloads are "
                           "measured at N=1 only")
        emit(args, res)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
