#!/usr/bin/env python3
"""Timing-only variants of the single-layer 3x3 kernel (results are garbage; only the duration is meaningful): the source
is patched textually, each variant becomes superresolution_amd/_exp/libconv_<name>.so, and `run` times them interleaved
in one process at 2160x3840 on the concat layer (32 + 64 -> 64, `Cfg<6,2,2>`) and the plain 64 -> 64 layer.
usage: exp_conv_variants.py build [names...] | run [rounds]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "superresolution_amd", "csrc")
OUT = os.path.join(ROOT, "superresolution_amd", "_exp")

DMA = [("                            issue_piece(IntC<TD>{}, dma_s0, dma_s1, dma_slot);", "")]
STORES = [("            buffer_store_b128_guarded(o, dst_srd, vox[k], store_soff);\n        };\n        auto epi_store2",
           "            asm volatile(\"\" :: \"v\"(o));\n        };\n        auto epi_store2")]
EPI = [("                constexpr bool EPI = !C::OUT_F32 && t >= 1 && KG < C::NST;", "                constexpr bool EPI = false;"),
       ("                        if constexpr (S < E) {\n                            constexpr int KD", "                        if constexpr (false && S < E) {\n                            constexpr int KD"),
       ("if constexpr (ml == 0 && !C::OUT_F32 && (t % 2 == 0) && (t / 2 < C::NST)) epi_read", "if constexpr (false) epi_read")]
STAGE = [("                    lds_write_acc_quad<ct, g>(wst ^ (unsigned)((ct * 8 + 2 * g) << 4), acc[ct]);", "")]
FRAG = [("                        bq[nxt][kc] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kc)]) ^ (unsigned)(C::kcl_of(kc) << 5));", ""),
        ("                            bq[nxt][kr] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kr)]) ^ (unsigned)(C::kcl_of(kr) << 5));", "")]
BARRIER = [("            __builtin_amdgcn_s_barrier();\n\n            const int y = y0 + i;", "\n            const int y = y0 + i;")]
NOPS = [('if constexpr (C::NCT == 2) asm volatile("s_nop 15\\n\\ts_nop 3"', 'if constexpr (C::NCT == 2) asm volatile(""'),
        ('else asm volatile("s_nop 15\\n\\ts_nop 3" : "+" SRHIP_ACC_REG0(acc[0]));', 'else asm volatile("" : "+" SRHIP_ACC_REG0(acc[0]));')]

VARIANTS = {
    "base": [],
    "no_dma": DMA,
    "no_stores": STORES,
    "no_epi": EPI,
    "no_stage": STAGE,
    "no_frag": FRAG,
    "no_barrier": BARRIER,
    "no_nops": NOPS,
    "mfma_only": DMA + EPI + STAGE + FRAG,
    "mfma_only_nobar": DMA + EPI + STAGE + FRAG + BARRIER + NOPS,
}


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv3x3_mfma.hip")).read()
    for name, patches in VARIANTS.items():
        if only and name not in only:
            continue
        text = src
        for a, b in patches:
            if text.count(a) < 1:
                raise SystemExit(f"variant {name}: pattern not found: {a[:70]!r}")
            text = text.replace(a, b)
        path = os.path.join(CSRC, f"_exp_conv_{name}.hip")
        with open(path, "w") as f:
            f.write(text)
        so = os.path.join(OUT, f"libconv_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950",
               "-shared", "-no-hip-rt", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable", path,
               os.path.join(CSRC, "runtime.hip"), "-o", so]
        print("building", name, flush=True)
        try:
            subprocess.check_call(cmd)
        finally:
            os.remove(path)


def run(rounds):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_
    from superresolution_amd._lib import Conv3x3Args
    H, W = 2160, 3840
    dev = torch.device("cuda")
    xa = torch.relu(torch.rand(1, H, W, 32, device=dev) - 0.5).to(torch.bfloat16)
    xb = torch.relu(torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16)
    wcat = F_.conv3x3_pack_weights(torch.randn(64, 96, 3, 3, device=dev) * 0.05, 32, 32, 64, 64)
    w64 = F_.conv3x3_pack_weights(torch.randn(64, 64, 3, 3, device=dev) * 0.05, 64, 64)
    bias = torch.randn(64, device=dev) * 0.1
    out = torch.empty(1, H, W, 64, dtype=torch.bfloat16, device=dev)
    cases = {
        "concat": Conv3x3Args(src_a=xa.data_ptr(), src_b=xb.data_ptr(), wpack=wcat.data_ptr(), bias=bias.data_ptr(), res=None,
                              dst=out.data_ptr(), n=1, h=H, w=W, ca=32, cb=64, cout=64, cout_stride=64, act=1, slope=0.0, alpha=1.0,
                              dst_nchw_f32=0, res_mode=0, ca_stride=32, cout_w=64, aux=None, aux_mode=0, dst2=None),
        "plain64": Conv3x3Args(src_a=xb.data_ptr(), src_b=None, wpack=w64.data_ptr(), bias=bias.data_ptr(), res=None,
                               dst=out.data_ptr(), n=1, h=H, w=W, ca=64, cb=0, cout=64, cout_stride=64, act=1, slope=0.0, alpha=1.0,
                               dst_nchw_f32=0, res_mode=0, ca_stride=64, cout_w=64, aux=None, aux_mode=0, dst2=None),
    }
    libs = {}
    names = sorted(f[len("libconv_"):-3] for f in os.listdir(OUT) if f.startswith("libconv_") and f.endswith(".so"))
    if "base" in names:
        names.remove("base")
        names.insert(0, "base")
    for name in names:
        lib = ctypes.CDLL(os.path.join(OUT, f"libconv_{name}.so"))
        lib.srhip_conv3x3_fwd.argtypes = [ctypes.POINTER(Conv3x3Args), ctypes.c_void_p]
        lib.srhip_conv3x3_fwd.restype = ctypes.c_int
        libs[name] = lib
    stream = torch.cuda.current_stream().cuda_stream
    for cname, a in cases.items():
        times = {n: [] for n in libs}
        for r in range(rounds + 1):
            for name, lib in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    assert lib.srhip_conv3x3_fwd(ctypes.byref(a), stream) == 0
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[name].append(e0.elapsed_time(e1) / 5)
        base = sorted(times["base"])[len(times["base"]) // 2]
        for name, ts in times.items():
            ts = sorted(ts)
            print(f"{cname:8s} {name:18s} median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  ({ts[len(ts) // 2] / base:.3f} x base)", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(only=sys.argv[2:] or None)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
