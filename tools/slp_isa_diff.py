#!/usr/bin/env python3
"""ISA of gradient_map_unshuffle4_nhwc_bf16_kernel (csrc/memops.hip) built with and without the SLP vectoriser:
registers, scratch, occupancy, instruction mix and every form of packed fp32 instruction (operand kinds and op_sel /
neg modifiers), next to the kernels that use packed fp32 by hand.  No GPU needed.  usage: slp_isa_diff.py [outdir]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else "/tmp/slp_isa"
os.makedirs(OUT, exist_ok=True)
KERNEL = "gradient_map_unshuffle4_nhwc_bf16_kernel"


def build(tag, flags):
    s = os.path.join(OUT, f"memops_{tag}.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *flags, "-S",
                        "--cuda-device-only", os.path.join(ROOT, "superresolution_amd", "csrc", "memops.hip"), "-o", s,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    return open(s).read().split("\n"), r.stderr


def bodies(lines, pat):
    for st in [i for i, l in enumerate(lines) if re.match(r"^_ZN.*" + pat + ".*:", l)]:
        e = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        yield lines[st].split(":")[0], lines[st:e]


def packed_forms(body):
    forms = collections.Counter()
    for l in body:
        m = re.match(r"\s+(v_pk_(?:add|mul|fma)_f32)\s+(.*)", l)
        if m:
            ops = m.group(2).split(";")[0]
            mod = " ".join(re.findall(r"(op_sel(?:_hi)?:\[[\d,]+\]|neg_lo:\[[\d,]+\]|neg_hi:\[[\d,]+\])", ops))
            kinds = tuple("v" if o.strip().startswith("v[") else ("s" if o.strip().startswith("s[") else "const") for o in ops.split(",")[1:4]
                          if not o.strip().startswith(("op_sel", "neg_")))
            forms[(m.group(1), kinds, mod)] += 1
    return forms


for tag, flags in (("slp", []), ("noslp", ["-fno-slp-vectorize"])):
    lines, remarks = build(tag, flags)
    blk = remarks[remarks.find(KERNEL):][:2500]
    res = {k: (re.search(k + r"[^:]*: (\S+)", blk) or [None, None])[1] for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs")}
    name, body = next(bodies(lines, KERNEL))
    ninstr = len([l for l in body if re.match(r"\s+[a-z]", l)])
    mix = collections.Counter(m.group(1) for l in body for m in [re.match(r"\s+(v_pk_\w+|v_fma_f32|v_sqrt_f32|s_waitcnt|s_barrier|ds_\w+|global_\w+)\b", l)] if m)
    print(f"== {KERNEL} built {'with' if tag == 'slp' else 'without'} the SLP vectoriser: {res}, {ninstr} instructions")
    print("   ", dict(sorted(mix.items())))
    for k, v in packed_forms(body).most_common():
        print("   ", v, k)
    if tag == "noslp":
        for pat in ("upsample_int_nhwc_kernelILi4", "layernorm_cf_bwd_dx_kernel"):
            for name, body in bodies(lines, pat):
                print(f"-- hand-packed {pat}:")
                for k, v in packed_forms(body).most_common():
                    print("   ", v, k)
