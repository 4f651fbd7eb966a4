#!/usr/bin/env python3
"""Timing-only variants of the weight-gradient kernel (results are garbage; only the duration is meaningful): the source
is patched textually, each variant becomes superresolution_amd/_exp/libwgrad_<name>.so, and `run` times them interleaved
in one process at 16 x 1024 x 1024, 64 -> 64.  usage: exp_wgrad_variants.py build | run [rounds]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "superresolution_amd", "csrc")
OUT = os.path.join(ROOT, "superresolution_amd", "_exp")

DMA = [("                    issue_piece(WIntC<decltype(m_c)::value + 0 * st>{}, dxs, dxslot, dzs, dzslot);", "")]
NOP = [('asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"', 'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"')]
BIAS = [("if (cit == sk % C::CIT) wg_mfma<9>", "if (false) wg_mfma<9>")]
XREAD = [("                    blo[nb] = lds_read_tr<(sn % C::NKS) * 16 * C::XPB>(xa[sn / C::NKS][0]);\n"
          "                    bhi[nb] = lds_read_tr<(sn % C::NKS) * 16 * C::XPB>(xa[sn / C::NKS][1]);\n", "")]
BARRIER = [("            else wg_vm_wait<C::TOP_WAIT>();\n            __builtin_amdgcn_s_barrier();", "            else wg_vm_wait<C::TOP_WAIT>();")]

# every DMA fetches the item's first rows again (served from cache): the issue cost without the HBM traffic
CACHED = [("            return in ? (unsigned)(y * W) * (unsigned)C::XPB : kOob;", "            return in ? (unsigned)((y0 + (r & 3)) * W) * (unsigned)C::XPB : kOob;"),
          ("return (r < nout) ? (unsigned)((y0 + r) * W) * (unsigned)C::ZPB : kOob; };", "return (r < nout) ? (unsigned)((y0 + (r & 3)) * W) * (unsigned)C::ZPB : kOob; };")]
ISSUE = ("                // the LDS-DMA pieces of the rows two iterations ahead, spread over the first steps\n"
         "                wg_static_for<st * PPS, ((st + 1) * PPS < NP ? (st + 1) * PPS : NP)>([&](auto m_c) {\n"
         "                    issue_piece(WIntC<decltype(m_c)::value + 0 * st>{}, dxs, dxslot, dzs, dzslot);\n"
         "                });\n")
MFMA3 = "                    wg_mfma<ky * 3 + kx>(acc[ky * 3 + kx], join(alo[sl][sk], ahi[sl][sk]), b);\n"
# the piece of a step right behind its first MFMA instead of behind its last
DMA_MID = [(ISSUE, ""), (MFMA3, MFMA3 + "                    if constexpr (ky == 0) { wg_static_for<st * PPS, ((st + 1) * PPS < NP ? (st + 1) * PPS : NP)>([&](auto m_c) {\n"
            "                        issue_piece(WIntC<decltype(m_c)::value + 0 * st + 0 * ky>{}, dxs, dxslot, dzs, dzslot); }); }\n")]
# all pieces of the row in one burst in front of the row's LDS reads
DMA_FRONT = [(ISSUE, ""), ("            // dz fragments of row r -> register slot PH\n",
             "            wg_static_for<0, C::MAXPX + C::MAXPZ>([&](auto m_c) { issue_piece(m_c, dxs, dxslot, dzs, dzslot); });\n"
             "            // dz fragments of row r -> register slot PH\n")]
# the pieces in the LAST steps of the row (the two last steps issue no LDS reads)
DMA_LATE = [("wg_static_for<st * PPS, ((st + 1) * PPS < NP ? (st + 1) * PPS : NP)>([&](auto m_c) {\n                    issue_piece(WIntC<decltype(m_c)::value + 0 * st>{}",
             "wg_static_for<(st >= NSTEP - NP ? st - (NSTEP - NP) : 0), (st >= NSTEP - NP ? st - (NSTEP - NP) + 1 : 0)>([&](auto m_c) {\n                    issue_piece(WIntC<decltype(m_c)::value + 0 * st>{}")]
# register-staged transport instead of LDS-DMA (timing only: the LDS write of a piece uses the registers the previous
# load of that piece slot targeted, without waiting for it): one buffer_load_dwordx4 + one ds_write_b128 per piece
REGS = [("        auto issue_piece = [&](auto m_c, const unsigned xsoff, const unsigned xslot, const unsigned zsoff, const unsigned zslot) {\n"
         "            constexpr int m = decltype(m_c)::value;\n",
         "        auto issue_piece = [&](auto m_c, const unsigned xsoff, const unsigned xslot, const unsigned zsoff, const unsigned zslot) {\n"
         "            constexpr int m = decltype(m_c)::value;\n"
         "            {\n"
         "                const bool isx = m < C::MAXPX;\n"
         "                const int mm = isx ? m : m - C::MAXPX;\n"
         "                const int qq = wave + mm * C::NWAVES;\n"
         "                const unsigned off = lds_base + (isx ? xslot : zslot) + (unsigned)qq * 1024u + (unsigned)lane * 16u;\n"
         "                const unsigned vo = isx ? xoff[mm < C::MAXPX ? mm : 0] : zoff[mm < C::MAXPZ ? mm : 0];\n"
         "                const unsigned so = isx ? xsoff : zsoff;\n"
         "                asm volatile(\"ds_write_b128 %0, %1\" :: \"v\"(off), \"v\"(stage_regs[m]) : \"memory\");\n"
         "                if (isx) asm volatile(\"buffer_load_dwordx4 %0, %1, %2, %3 offen\" : \"=v\"(stage_regs[m]) : \"v\"(vo), \"s\"(rsrc_x), \"s\"(so));\n"
         "                else asm volatile(\"buffer_load_dwordx4 %0, %1, %2, %3 offen\" : \"=v\"(stage_regs[m]) : \"v\"(vo), \"s\"(rsrc_z), \"s\"(so));\n"
         "                return;\n"
         "            }\n"),
        ("    bf16x4 alo[3][C::NKS], ahi[3][C::NKS];   // dz fragments of the three newest rows\n",
         "    bf16x4 alo[3][C::NKS], ahi[3][C::NKS];   // dz fragments of the three newest rows\n"
         "    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));\n"
         "    u32x4_t stage_regs[C::MAXPX + C::MAXPZ];\n"
         "    for (int i = 0; i < C::MAXPX + C::MAXPZ; ++i) stage_regs[i] = u32x4_t{0u, 0u, 0u, 0u};\n")]
VARIANTS = {"dma_regs": REGS, "dma_mid": DMA_MID, "dma_front": DMA_FRONT, "dma_late": DMA_LATE, "dma_cached": CACHED, "base": [], "no_dma": DMA, "no_nop": NOP, "no_bias": BIAS, "no_xread": XREAD, "no_barrier": BARRIER,
            "mfma_only": DMA + XREAD + BIAS}


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv3x3_wgrad.hip")).read()
    for name, patches in VARIANTS.items():
        if only and name not in only:
            continue
        text = src
        for a, b in patches:
            if text.count(a) < 1:
                raise SystemExit(f"variant {name}: pattern not found: {a[:60]!r}")
            text = text.replace(a, b)
        path = os.path.join(CSRC, f"_exp_wgrad_{name}.hip")
        with open(path, "w") as f:
            f.write(text)
        so = os.path.join(OUT, f"libwgrad_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950",
               "-shared", "-no-hip-rt", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable", path,
               os.path.join(CSRC, "runtime.hip"), "-o", so]
        print("building", name, flush=True)
        subprocess.check_call(cmd)
        os.remove(path)


def run(rounds):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_   # noqa: F401  (loads the HIP runtime the variants bind to)
    from superresolution_amd import _lib
    n, h, w = 16, 1024, 1024
    dev = torch.device("cuda")
    x = torch.relu(torch.rand(n, h, w, 64, device=dev) - 0.5).to(torch.bfloat16)
    dz = ((torch.rand(n, h, w, 64, device=dev) - 0.5) * (torch.rand(n, h, w, 64, device=dev) > 0.5)).to(torch.bfloat16)
    dw = torch.empty(64, 64, 3, 3, device=dev)
    db = torch.empty(64, device=dev)
    ws = torch.empty(_lib.lib.srhip_conv3x3_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
    libs = {}
    for f in sorted(os.listdir(OUT)):
        if f.startswith("libwgrad_") and f.endswith(".so"):
            lib = ctypes.CDLL(os.path.join(OUT, f))
            lib.srhip_conv3x3_wgrad.restype, lib.srhip_conv3x3_wgrad.argtypes = _lib._SIGS["srhip_conv3x3_wgrad"] if hasattr(_lib, "_SIGS") else (
                ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 10 + [ctypes.c_float, ctypes.c_void_p])
            libs[f[len("libwgrad_"):-3]] = lib
    stream = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in libs}
    for r in range(rounds + 1):
        for name, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert lib.srhip_conv3x3_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), n, h, w, 64, 64, 0,
                                               64, 64, 64, 0, 1.0, stream) == 0
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / 5)
    base = sorted(times["base"])[len(times["base"]) // 2]
    for name, ts in times.items():
        ts = sorted(ts)
        print(f"{name:14s} median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  ({ts[len(ts) // 2] / base:.3f} x base)")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(only=sys.argv[2:] or None)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
