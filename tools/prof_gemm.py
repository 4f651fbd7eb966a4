#!/usr/bin/env python3
"""A few launches of the implicit-GEMM convolution on temporal-network layer shapes, for rocprofv3 counter passes:
    rocprofv3 --pmc <counters> -- python3 tools/prof_gemm.py [case]
cases: wide (160 -> 128 3x3 @544x960), k1 (160 -> 160 1x1 @544x960), deep (256 -> 256 3x3 @68x120), s2 (64 -> 128 3x3 stride 2 @544x960)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd import functional as F_  # noqa: E402

CASES = {"wide": (160, 128, 3, 1, 544, 960), "k1": (160, 160, 1, 1, 544, 960), "deep": (256, 256, 3, 1, 68, 120),
         "s2": (64, 128, 3, 2, 544, 960), "mid": (128, 128, 3, 1, 272, 480), "q192": (192, 192, 3, 1, 136, 240),
         "q384": (384, 256, 3, 1, 136, 240), "deep192": (256, 192, 3, 1, 68, 120), "s2q": (192, 256, 3, 2, 136, 240),
         "wide160": (160, 160, 3, 1, 544, 960), "w112": (112, 112, 3, 1, 544, 960), "w128_64": (128, 64, 3, 1, 544, 960),
         "gate": (96, 2, 3, 1, 544, 960), "n96_64": (96, 64, 3, 1, 544, 960), "n128_48": (128, 48, 3, 1, 544, 960)}


def main():
    names = sys.argv[1:] or ["wide"]
    g = torch.Generator().manual_seed(0)
    for name in names:
        cin, cout, k, s, h, w = CASES[name]
        x = (torch.rand(1, h, w, cin, generator=g) - 0.5).to(torch.bfloat16).cuda()
        wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
        wp = F_.conv_gemm_pack_weights(wt, cin)
        for _ in range(3):
            F_.conv_gemm(x, wp, None, cout, k, stride=s, padding=k // 2, act=F_.ACT_RELU)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in evs:
            a.record(); F_.conv_gemm(x, wp, None, cout, k, stride=s, padding=k // 2, act=F_.ACT_RELU); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        flop = 2.0 * ho * wo * cin * cout * k * k
        print(f"{name}: {cin}->{cout} k{k} s{s} @{h}x{w}: median {ts[5] * 1e3:.1f} us = {flop / ts[5] / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
