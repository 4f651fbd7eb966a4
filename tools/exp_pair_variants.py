#!/usr/bin/env python3
"""Timing-only variants of the fused pair kernel (results are garbage; only the duration is meaningful): the source is
patched textually, each variant becomes build/exp/libpair_<name>.so, and `run` times them interleaved in one process
at 2160x3840.  usage: exp_pair_variants.py build | run [rounds]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "superresolution_amd", "csrc")
OUT = os.path.join(ROOT, "superresolution_amd", "_exp")

L1_EPI = ("                    if constexpr (t < 8) {\n                        constexpr int qct", "                    if constexpr (false && t < 8) {\n                        constexpr int qct")
L2_EPI = [("                constexpr bool EPI = t >= 2;", "                constexpr bool EPI = false && t >= 2;"),
          ("if constexpr (ml == 0 && (t % 2 == 1) && (t / 2 < P::NST)) epi_read", "if constexpr (false && ml == 0 && (t % 2 == 1) && (t / 2 < P::NST)) epi_read"),
          ("if constexpr (ml == 0 && (t % 2 == 0) && (t / 2 < NRES)) resv", "if constexpr (false && ml == 0 && (t % 2 == 0) && (t / 2 < NRES)) resv")]
FRAG = [("                        bq[nxt][kr] = lds_read_b128_sel<IMM, ((Q + kyn) & (P::RING - 1)) * P::SLOT_BYTES>(rowx[kyn], rd[tn % 3][kr]);", ""),
        ("                        bq[nxt][kr] = lds_read_b128_off<((Q + kyn) & (P::NY - 1)) * P::SLOT_BYTES>(rd[tn % 3][kr]);", "")]
STAGE = [("if constexpr (t == 0 && ml < 4) lds_write_acc_quad", "if constexpr (false && t == 0 && ml < 4) lds_write_acc_quad"),
         ("if constexpr (t == 8 && ml >= 4) lds_write_acc_quad", "if constexpr (false && t == 8 && ml >= 4) lds_write_acc_quad")]
DMA = [("if constexpr (t < P::MAXP && ml == 6) issue_piece(IntC<t + 0 * ml>{}, dma_soff, dma_slot);", "")]
BARRIER = [("            __builtin_amdgcn_s_barrier();\n\n            // mid row s-1", "\n            // mid row s-1"),
           ("                    __builtin_amdgcn_s_barrier();\n                    if (s + Q >= P::LAG) row_body(q_c, s + Q - P::LAG);\n                }\n            });\n        }\n        // drain",
            "                    if (s + Q >= P::LAG) row_body(q_c, s + Q - P::LAG);\n                }\n            });\n        }\n        // drain")]
NOPS = [('asm volatile("s_nop 15\\n\\ts_nop 3" : "+v"(acc[PAR][0]), "+v"(acc[PAR][1]));', 'asm volatile("" : "+v"(acc[PAR][0]), "+v"(acc[PAR][1]));')]
STORES = [("            buffer_store_b128_guarded(o, dst_srd, vox[k], store_soff);", "            asm volatile(\"\" :: \"v\"(o));")]

VARIANTS = {
    "base": [],
    "no_l1epi": [L1_EPI],
    "no_l2epi": L2_EPI,
    "no_frag": FRAG,
    "no_stage": STAGE,
    "no_dma": DMA,
    "no_barrier": BARRIER,
    "no_stores": STORES,
    "no_epis": [L1_EPI] + L2_EPI + STAGE,
    "mfma_only": [L1_EPI] + L2_EPI + FRAG + STAGE + DMA,
    "mfma_only_nobar": [L1_EPI] + L2_EPI + FRAG + STAGE + DMA + BARRIER,
}


def build(only=None, rev=None):
    """only: names to build (default all).  rev: also build `rev_<rev>` from that git revision of the source (an older
    version of the kernel for an A/B in one process)."""
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv3x3_pair.hip")).read()
    todo = dict(VARIANTS)
    if only:
        todo = {k: v for k, v in todo.items() if k in only}
    if rev:
        old = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:superresolution_amd/csrc/conv3x3_pair.hip"], text=True)
        todo[f"rev_{rev}"] = old
    flags = {}
    for name in list(todo):
        if os.environ.get("EXP_SLP"):       # the same variant with the SLP vectoriser left on (packed fp32 beside MFMAs)
            todo[name + "_slp"] = todo[name]
            flags[name + "_slp"] = ["-fslp-vectorize"]
    for name, patches in todo.items():
        text = src
        if isinstance(patches, str):
            text, patches = patches, []
        for a, b in patches:
            if text.count(a) < 1:
                raise SystemExit(f"variant {name}: pattern not found: {a[:60]!r}")
            text = text.replace(a, b)
        path = os.path.join(CSRC, f"_exp_pair_{name}.hip")
        with open(path, "w") as f:
            f.write(text)
        so = os.path.join(OUT, f"libpair_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-shared",
               "-no-hip-rt", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable", *flags.get(name, []), path, os.path.join(CSRC, "runtime.hip"), "-o", so]
        print("building", name, flush=True)
        subprocess.check_call(cmd)
        os.remove(path)


def run(rounds, only=None):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_
    from superresolution_amd._lib import Conv3x3PairArgs
    H, W = 2160, 3840
    dev = torch.device("cuda")
    x = (torch.rand(1, H, W, 64, device=dev) - 0.5).to(torch.bfloat16)
    wp1 = F_.conv3x3_pack_weights(torch.randn(64, 64, 3, 3, device=dev) * 0.05, 64, 64)
    wp2 = F_.conv3x3_pack_weights(torch.randn(64, 64, 3, 3, device=dev) * 0.05, 64, 64)
    b1 = torch.randn(64, device=dev) * 0.1
    b2 = torch.randn(64, device=dev) * 0.1
    out = torch.empty_like(x)
    libs = {}
    names = sorted(f[len("libpair_"):-3] for f in os.listdir(OUT) if f.startswith("libpair_") and f.endswith(".so"))
    names = [n for n in names if only is None or n in only]
    if "base" in names:
        names.remove("base")
        names.insert(0, "base")
    for name in names:
        so = os.path.join(OUT, f"libpair_{name}.so")
        if os.path.exists(so):
            lib = ctypes.CDLL(so)
            lib.srhip_conv3x3_pair_fwd.argtypes = [ctypes.POINTER(Conv3x3PairArgs), ctypes.c_void_p]
            lib.srhip_conv3x3_pair_fwd.restype = ctypes.c_int
            libs[name] = lib
    a = Conv3x3PairArgs(src=x.data_ptr(), wpack1=wp1.data_ptr(), bias1=b1.data_ptr(), wpack2=wp2.data_ptr(), bias2=b2.data_ptr(),
                        mid=None, dst=out.data_ptr(), n=1, h=H, w=W, act2=0, slope2=0.0, alpha=0.2)
    stream = torch.cuda.current_stream().cuda_stream
    times = {n: [] for n in libs}
    for r in range(rounds + 1):
        for name, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert lib.srhip_conv3x3_pair_fwd(ctypes.byref(a), stream) == 0
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / 5)
    base = sorted(times["base"])[len(times["base"]) // 2]
    for name, ts in times.items():
        ts = sorted(ts)
        print(f"{name:18s} median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  ({ts[len(ts) // 2] / base:.3f} x base)")


if __name__ == "__main__":
    # build [--rev REV] [names...] | run [rounds] [names...]
    if sys.argv[1] == "build":
        args = sys.argv[2:]
        rev = None
        if args and args[0] == "--rev":
            rev, args = args[1], args[2:]
        build(only=args or None, rev=rev)
    else:
        args = sys.argv[2:]
        rounds = int(args[0]) if args else 7
        run(rounds, only=args[1:] or None)
