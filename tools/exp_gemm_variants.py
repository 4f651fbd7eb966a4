#!/usr/bin/env python3
"""Timing-only variants of the implicit-GEMM convolution (results are garbage; only the duration is meaningful): the
source is patched textually, each variant becomes superresolution_amd/_exp/libgemm_<name>.so, and `run` times them
interleaved in one process on temporal-network layer shapes.
usage: exp_gemm_variants.py build [names...] | run [rounds]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "superresolution_amd", "csrc")
OUT = os.path.join(ROOT, "superresolution_amd", "_exp")

BAR = [("            __syncthreads();   // everyone is done reading the previous stage\n", ""),
       ("            });\n            __syncthreads();\n        }\n        {", "            });\n        }\n        {")]
COMMIT = [("if (C::XP1 % NTHREADS == 0 || q < C::XP1) *(uint4*)", "if (tid < 0) *(uint4*)")]
FRAG = [("                    bq[j & 1][pt] = g_lds_read<boff + pt * 32 * 16>(baddr);", ""),
        ("                    af[j & 1][ct] = g_lds_read<(j * NT + ct) * 1024>(aaddr);", "")]
MFMA = [("                    acc[pt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i & 1][ct], bq[i & 1][pt], acc[pt][ct], 0, 0, 0);",
         "                    acc[pt][ct][0] += (float)af[i & 1][ct][0] + (float)bq[i & 1][pt][0];")]

VARIANTS = {
    "base": ([], 0),
    "no_loads": ([], 3),
    "no_loads_bar": (BAR, 3),
    "no_loads_commit": (COMMIT, 3),
    "no_loads_bar_commit": (BAR + COMMIT, 3),
    "no_loads_frag": (FRAG, 3),
    "mfma_only": (BAR + COMMIT + FRAG, 3),
    "mfma_only_nostore": (BAR + COMMIT + FRAG, 7),
    "no_mfma": (MFMA, 0),
    "no_bar": (BAR, 0),
}


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv_gemm.hip")).read()
    for name, (patches, exp) in VARIANTS.items():
        if only and name not in only:
            continue
        text = src
        for a, b in patches:
            if text.count(a) < 1:
                raise SystemExit(f"variant {name}: pattern not found: {a[:70]!r}")
            text = text.replace(a, b)
        path = os.path.join(CSRC, f"_exp_gemm_{name}.hip")
        with open(path, "w") as f:
            f.write(text)
        so = os.path.join(OUT, f"libgemm_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950",
               "-shared", "-no-hip-rt", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable", f"-DGEMM_EXP={exp}",
               path, os.path.join(CSRC, "runtime.hip"), "-o", so]
        print("building", name, flush=True)
        try:
            subprocess.check_call(cmd)
        finally:
            os.remove(path)


CASES = {"wide": (160, 128, 3, 1, 544, 960), "wide160": (160, 160, 3, 1, 544, 960), "mid": (128, 128, 3, 1, 272, 480),
         "w128_64": (128, 64, 3, 1, 544, 960), "q384": (384, 256, 3, 1, 136, 240)}


def run(rounds):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_
    from superresolution_amd._lib import ConvGemmArgs
    dev = torch.device("cuda")
    libs = {}
    names = sorted(f[len("libgemm_"):-3] for f in os.listdir(OUT) if f.startswith("libgemm_") and f.endswith(".so"))
    if "base" in names:
        names.remove("base")
        names.insert(0, "base")
    for name in names:
        lib = ctypes.CDLL(os.path.join(OUT, f"libgemm_{name}.so"))
        lib.srhip_conv_gemm_fwd.argtypes = [ctypes.POINTER(ConvGemmArgs), ctypes.c_void_p]
        lib.srhip_conv_gemm_fwd.restype = ctypes.c_int
        libs[name] = lib
    stream = torch.cuda.current_stream().cuda_stream
    for cname, (cin, cout, k, s, h, w) in CASES.items():
        x = (torch.rand(1, h, w, cin, device=dev) - 0.5).to(torch.bfloat16)
        wp = F_.conv_gemm_pack_weights(torch.randn(cout, cin, k, k, device=dev) * 0.05, cin)
        bias = torch.randn(cout, device=dev) * 0.1
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        out = torch.empty(1, ho, wo, F_.pad16(cout), dtype=torch.bfloat16, device=dev)
        a = ConvGemmArgs()
        a.src[0] = x.data_ptr(); a.c[0] = cin; a.cstride[0] = cin; a.nsrc = 1
        a.wpack = wp.data_ptr(); a.bias = bias.data_ptr(); a.res = None; a.dst = out.data_ptr()
        a.n, a.h, a.w, a.cout, a.cout_stride = 1, h, w, cout, F_.pad16(cout)
        a.ksize, a.stride, a.pad, a.act, a.slope, a.alpha, a.res_mode = k, s, k // 2, 1, 0.0, 1.0, 0
        flop = 2.0 * ho * wo * cin * cout * k * k
        times = {n: [] for n in libs}
        for r in range(rounds + 1):
            for name, lib in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    assert lib.srhip_conv_gemm_fwd(ctypes.byref(a), stream) == 0
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[name].append(e0.elapsed_time(e1) / 5)
        base = sorted(times["base"])[len(times["base"]) // 2]
        for name, ts in times.items():
            ts = sorted(ts)
            med = ts[len(ts) // 2]
            print(f"{cname:8s} {name:22s} median {med * 1e3:7.1f} us  ({med / base:.3f} x base, {flop / med / 1e9:6.0f} TFLOP/s nominal)", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(only=sys.argv[2:] or None)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
