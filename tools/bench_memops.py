#!/usr/bin/env python3
"""Micro-benchmarks of the HBM-bound operators at the 4K configuration (dev helper, GPU box).
Prints algorithmic GB/s (compulsory reads + writes) per operator; peak HBM is ~8 TB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import functional as F_

dev = torch.device("cuda")
H, W, h, w = 2160, 3840, 540, 960


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def headline():
    """The two operator figures BASELINE.json's configs ask for next to the frame rate (config 5: warp and pixel
    shuffle at the 4K shapes) -> {"warp_gbps", "warp_smooth_motion_gbps", "pixel_shuffle_gbps", "pixel_shuffle_head_gbps"}
    (algorithmic GB/s, median of 10 launches each)."""
    f32 = lambda *s: torch.rand(*s, device=dev)
    fr, mv = f32(1, 3, H, W), (torch.randn(1, 2, H, W, device=dev) * 4).clamp(-32, 32)
    smooth = torch.nn.functional.interpolate(torch.randn(1, 2, 28, 48, device=dev) * 4, size=(H, W), mode="bilinear",
                                             align_corners=True).clamp(-32, 32).contiguous()
    nb = (2 * 3 + 2) * H * W * 4
    out = {"warp_gbps": nb / timeit(lambda: F_.backward_warp(fr, mv, "bilinear", True)) / 1e6,
           "warp_smooth_motion_gbps": nb / timeit(lambda: F_.backward_warp(fr, smooth, "bilinear", True)) / 1e6}
    x = f32(1, 48, h, w)
    out["pixel_shuffle_gbps"] = 2 * 48 * h * w * 4 / timeit(lambda: F_.pixel_shuffle(x, 4)) / 1e6
    head = (torch.rand(1, h, w, 48, device=dev) - 0.5).to(torch.bfloat16)
    out["pixel_shuffle_head_gbps"] = (48 * h * w * 2 + 3 * H * W * 4) / timeit(lambda: F_.pixel_shuffle_nhwc_to_nchw_f32(head, 3, 4)) / 1e6
    return out


def suite(verbose=True):
    """-> list of {"op", "ms", "bytes", "gbps", "frac_of_hbm_peak"} (median of 10 launches each)."""
    results = []

    def line(name, ms, nbytes):
        results.append({"op": name, "ms": ms, "bytes": nbytes, "gbps": nbytes / ms / 1e6, "frac_of_hbm_peak": nbytes / ms / 1e6 / 8000.0})
        if verbose:
            print(f"{name:52s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000.0:5.1%} of 8 TB/s)", flush=True)


    bf = lambda *s: (torch.rand(*s, device=dev) - 0.5).to(torch.bfloat16)
    f32 = lambda *s: torch.rand(*s, device=dev)

    # prev SR output warped by HR motion vectors (config C4): 3 channels fp32 NCHW + 2-channel motion
    fr, mv = f32(1, 3, H, W), (torch.randn(1, 2, H, W, device=dev) * 4).clamp(-32, 32)
    line("warp bilinear f32 NCHW 3ch @4K", timeit(lambda: F_.backward_warp(fr, mv, "bilinear", True)), (2 * 3 + 2) * H * W * 4)
    # the same launch with spatially smooth motion (a rendered frame's vectors vary slowly; independent per-pixel vectors
    # send every lane of a wave to a different cache line)
    smooth = torch.nn.functional.interpolate(torch.randn(1, 2, 28, 48, device=dev) * 4, size=(H, W), mode="bilinear",
                                             align_corners=True).clamp(-32, 32).contiguous()
    line("warp bilinear f32 NCHW 3ch @4K, smooth motion", timeit(lambda: F_.backward_warp(fr, smooth, "bilinear", True)),
         (2 * 3 + 2) * H * W * 4)
    # the same warp fused with the trunk-input build of the recurrent network: 10 G-buffer planes + the warped 3-channel
    # frame -> one 16-channel NHWC bf16 pixel (40 + 8 + 12 B read, 32 B written per pixel)
    gb = f32(1, 10, H, W)
    line("warp + concat + layout -> NHWC bf16 16ch @4K", timeit(lambda: F_.warp_concat_nhwc(gb, fr, mv, "bilinear", True)), (10 + 2 + 3) * H * W * 4 + 16 * H * W * 2)
    line("warp + concat + layout -> NHWC bf16 16ch @4K, smooth motion", timeit(lambda: F_.warp_concat_nhwc(gb, fr, smooth, "bilinear", True)),
         (10 + 2 + 3) * H * W * 4 + 16 * H * W * 2)
    del gb
    fr16 = bf(1, H, W, 16)
    line("warp bilinear bf16 NHWC 16ch @4K", timeit(lambda: F_.backward_warp_nhwc(fr16, mv, "bilinear", True)), 2 * 16 * H * W * 2 + 2 * H * W * 4)
    line("warp bilinear bf16 NHWC 16ch @4K, smooth motion", timeit(lambda: F_.backward_warp_nhwc(fr16, smooth, "bilinear", True)),
         2 * 16 * H * W * 2 + 2 * H * W * 4)
    # reference-shaped LR warp: 13 channels (frames + buffers) at 540x960
    fl, ml = bf(1, h, w, 16), torch.randn(1, 2, h, w, device=dev)
    line("warp bilinear bf16 NHWC 16ch @LR", timeit(lambda: F_.backward_warp_nhwc(fl, ml, "bilinear", True)), 2 * 16 * h * w * 2 + 2 * h * w * 4)
    # network head: 48-channel LR map -> 3-channel 4K image
    head = bf(1, h, w, 48)
    line("pixel shuffle head NHWC bf16 -> NCHW f32 @4K", timeit(lambda: F_.pixel_shuffle_nhwc_to_nchw_f32(head, 3, 4)), 48 * h * w * 2 + 3 * H * W * 4)
    x = f32(1, 48, h, w)
    line("pixel shuffle f32 NCHW r=4", timeit(lambda: F_.pixel_shuffle(x, 4)), 2 * 48 * h * w * 4)
    g = f32(1, 10, H, W)
    line("pixel unshuffle f32 NCHW 10ch r=4", timeit(lambda: F_.pixel_unshuffle(g, 4)), 2 * 10 * H * W * 4)
    fm = bf(1, h, w, 32)
    line("upsample x4 bf16 NHWC 32ch -> 4K", timeit(lambda: F_.upsample_grid_bilinear_nhwc(fm, H, W)), 32 * (h * w + H * W) * 2)
    line("layout f32 NCHW 10ch -> bf16 NHWC 16ch @4K", timeit(lambda: F_.nchw_f32_to_nhwc_bf16(g)), H * W * (10 * 4 + 16 * 2))
    a, b = bf(1, H, W, 32), bf(1, H, W, 64)
    line("concat 32+64 bf16 NHWC @4K", timeit(lambda: F_.concat_nhwc(a, b)), 2 * 96 * H * W * 2)
    line("nearest down x4 f32 10ch from 4K", timeit(lambda: F_.nearest_down(g, 0.25)), 2 * 10 * h * w * 4)
    y = bf(1, H, W, 64)
    line("relu mask bf16 64ch @4K", timeit(lambda: F_.relu_mask(b, y)), 3 * 64 * H * W * 2)
    # channels-first LayerNorm (core/arch/CAMixerSR.py:29-37) on a 64-channel fp32 NCHW map at LR, forward and backward
    xl, wl, bl = f32(1, 64, h, w), f32(64), f32(64)
    line("layernorm_cf fwd f32 64ch @LR", timeit(lambda: F_.layernorm_cf(xl, wl, bl)), 2 * 64 * h * w * 4)
    dyl = f32(1, 64, h, w)
    line("layernorm_cf bwd f32 64ch @LR", timeit(lambda: F_.layernorm_cf_bwd(xl, wl, dyl)), 3 * 64 * h * w * 4)
    # backward of the fixed-ratio up-sampling (gather form, ReLU of the producer fused) and of the fp32 warp (scatter-add)
    dup = bf(1, H, W, 32)
    line("upsample x4 bwd bf16 NHWC 32ch 4K -> LR", timeit(lambda: F_.upsample_grid_bilinear_bwd_nhwc(dup, h, w, relu_out=fm)), 32 * (h * w * 2 + H * W) * 2)
    frl, mvl, dyw = f32(1, 3, h, w), torch.randn(1, 2, h, w, device=dev), f32(1, 3, h, w)
    line("warp bilinear bwd f32 NCHW 3ch @LR", timeit(lambda: F_.backward_warp_bwd(frl, mvl, dyw, True)), (3 * 3 + 2 + 2) * h * w * 4)
    sr, tg = f32(1, 3, H, W), f32(1, 3, H, W)
    line("fused L1+SSIM fwd+bwd 3ch @4K", timeit(lambda: F_.l1_ssim_loss(sr, tg, 1.0, 0.5, 0.0, 10.0)), 11 * 3 * H * W * 4)

    return results


if __name__ == "__main__":
    suite()
