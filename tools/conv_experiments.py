#!/usr/bin/env python3
"""Builds timing-only variants of the 3x3 MFMA kernel (results are WRONG by construction) to see what
each part of a row costs.  Variants are text patches of conv3x3_mfma.hip compiled into
superresolution_amd/_exp/libsrhip_<name>.so; run with SRHIP_LIBRARY=<that file> tools/bench_conv.py quick.
usage: conv_experiments.py build | list"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "superresolution_amd", "csrc")
OUT = os.path.join(ROOT, "superresolution_amd", "_exp")
BUILD = os.path.join(ROOT, "build", "exp")

ONLY_64 = [("    SRHIP_CONV_CASE(1, 1, 1)\n    SRHIP_CONV_CASE(2, 1, 1)\n    SRHIP_CONV_CASE(3, 1, 1)\n    SRHIP_CONV_CASE(4, 1, 1)\n"
            "    SRHIP_CONV_CASE(6, 1, 1)\n    SRHIP_CONV_CASE(1, 2, 1)\n    SRHIP_CONV_CASE(2, 2, 1)\n    SRHIP_CONV_CASE(3, 2, 1)\n",
            ""),
           ("    SRHIP_CONV_CASE(6, 2, 2)\n", ""),
           ("        if (a->dst_nchw_f32) return launch_conv<Cfg<KC, CT, CSPLIT, true, 0>>(p, s);                    \\\n", ""),
           ("        if (a->res != nullptr && a->res_mode == SRHIP_RES_MASK) return launch_conv<Cfg<KC, CT, CSPLIT, false, 2>>(p, s); \\\n", ""),
           ("        if (a->res != nullptr) return launch_conv<Cfg<KC, CT, CSPLIT, false, 1>>(p, s);                 \\\n", "")]

NO_BARRIER = [("            __builtin_amdgcn_s_barrier();\n\n            const int y = y0 + i;", "            const int y = y0 + i;")]
NO_EPI2 = [("constexpr bool EPI = !C::OUT_F32 && t >= 1 && KG < C::NST;", "constexpr bool EPI = false && KG < C::NST;")]
NO_DMA = [("                            issue_piece(IntC<TD>{}, dma_s0, dma_s1, dma_slot);", "                            (void)dma_s0; (void)dma_s1; (void)dma_slot;"),
          ("            if (i == 0) {\n                vm_wait<0>();\n            } else if (i < C::LA - 2) {", "            if (true) {\n            } else if (i < C::LA - 2) {")]
DMA_CACHED = [("            soff0 = yin ? pix * (unsigned)(csrc0 * 2) : kOobOffset;", "            soff0 = yin ? (pix & 0xfffu) * (unsigned)(csrc0 * 2) : kOobOffset;")]
NO_STORE = [("            __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 0);", "            if (store_soff == 12345u) __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 0);")]
NT_STORE = [("            __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 0);", "            __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 2);")]
NT_LOAD = [("    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);", "    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 2);")]
SC_STORE = [("            __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 0);", "            __builtin_amdgcn_raw_buffer_store_b128(o, dst_rsrc, vox[k], store_soff, 17);")]
LA6 = [("    static constexpr int LA = 5;", "    static constexpr int LA = 6;")]
LA4 = [("    static constexpr int LA = 5;", "    static constexpr int LA = 4;")]
NO_FRAG = [("                        bq[nxt][kc] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kc)]) ^ (unsigned)(C::kcl_of(kc) << 5));", "                        (void)kyn; (void)tn;"),
           ("                                bq[nxt][kr] = lds_read_b128((rowa[kyn] + rd[tn % 3][C::sub_of(kr)]) ^ (unsigned)(C::kcl_of(kr) << 5));", "                                (void)kr;")]
NO_EPI1 = [("                    lds_write_acc_quad<ct, g>(wst ^ (unsigned)((ct * 8 + 2 * g) << 4), acc[ct]);", "                    (void)wst;")]
NO_NOP = [('asm volatile("s_nop 15\\n\\ts_nop 3" : "+" SRHIP_ACC_REG0(acc[0]), "+" SRHIP_ACC_REG1(acc[1]));', 'asm volatile("" : "+" SRHIP_ACC_REG0(acc[0]), "+" SRHIP_ACC_REG1(acc[1]));')]

VARIANTS = {
    "base": [],
    "nobarrier": NO_BARRIER,
    "noepi2": NO_EPI2,
    "nodma": NO_DMA,
    "dmacached": DMA_CACHED,
    "nostore": NO_STORE,
    "la6": LA6,
    "la4": LA4,
    "la6nt": LA6 + NT_STORE + NT_LOAD,
    "ntstore": NT_STORE,
    "ntload": NT_LOAD,
    "ntboth": NT_STORE + NT_LOAD,
    "scstore": SC_STORE,
    "nostore_dmacached": NO_STORE + DMA_CACHED,
    "nofrag": NO_FRAG,
    "noepi": NO_EPI2 + NO_EPI1 + NO_NOP,
    "mfmaonly": NO_BARRIER + NO_EPI2 + NO_DMA + NO_FRAG + NO_EPI1 + NO_NOP,
    "mfma_frag": NO_BARRIER + NO_EPI2 + NO_DMA + NO_EPI1 + NO_NOP,
}


def patch(text, subs, name):
    for old, new in subs:
        if old not in text:
            raise SystemExit(f"variant {name}: anchor not found:\n{old}")
        text = text.replace(old, new)
    return text


def build():
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(BUILD, exist_ok=True)
    base = open(os.path.join(SRC, "conv3x3_mfma.hip")).read()
    objs = [os.path.join(ROOT, "build", "obj", o) for o in ("runtime.o", "conv3x3_wgrad.o", "conv_direct.o", "memops.o", "loss.o")]
    procs = []
    only = sys.argv[2:]
    for name, subs in VARIANTS.items():
        if only and name not in only:
            continue
        src = os.path.join(BUILD, f"conv_{name}.hip")
        text = patch(base, ONLY_64 + subs, name).replace('#include "srhip_common.h"', f'#include "{SRC}/srhip_common.h"')
        open(src, "w").write(text)
        obj = os.path.join(BUILD, f"conv_{name}.o")
        lib = os.path.join(OUT, f"libsrhip_{name}.so")
        cmd = (f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-value -c {src} -o {obj} && "
               f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o {lib} {obj} {' '.join(objs)}")
        procs.append((name, subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) % 4 == 0:
            for n, p in procs[-4:]:
                out, _ = p.communicate()
                print(n, "rc", p.returncode, out[-600:] if p.returncode else "")
    for n, p in procs[len(procs) // 4 * 4:]:
        out, _ = p.communicate()
        print(n, "rc", p.returncode, out[-600:] if p.returncode else "")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "list":
        print(" ".join(v for v in VARIANTS if os.path.exists(os.path.join(OUT, f"libsrhip_{v}.so"))))
    else:
        build()
