"""The hand-scheduled kernels issue loads from inline asm, which the compiler's wait bookkeeping cannot see.
tools/isa_lint.py compiles them to gfx950 assembly and proves, on every control-flow path, that nothing reads
or overwrites a load's destination registers before the fence behind its hand-placed wait.  (No GPU needed:
hipcc cross-compiles.)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "tools", "isa_lint.py")
KERNELS = ["conv3x3_mfma.hip", "conv3x3_wgrad.hip", "conv3x3_pair.hip", "conv_gemm.hip", "conv1x1_chain.hip"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_register_touched_between_asm_load_and_fence():
    def run(name):
        return subprocess.run([sys.executable, LINT, os.path.join(ROOT, "superresolution_amd", "csrc", name)],
                              capture_output=True, text=True, timeout=900)
    with ThreadPoolExecutor(2) as ex:
        results = list(ex.map(run, KERNELS))
    for name, r in zip(KERNELS, results):
        assert r.returncode == 0, f"{name}:\n{r.stdout[-3000:]}\n{r.stderr[-2000:]}"
        assert " 0 violation(s)" in r.stdout
        checked = int(r.stdout.strip().splitlines()[-1].split(":")[1].split()[0])
        assert checked > 100, "the lint found no asm loads: the assembly patterns it looks for have changed"


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_compiler_formed_packed_fp32_in_the_vector_alu_kernels(tmp_path):
    """A kernel whose fp32 arithmetic the SLP vectoriser had packed (v_pk_add_f32 / v_pk_mul_f32) stopped repeating bit
    for bit when a second process shared the GPU; built scalar it repeats.  Every HIP file is therefore built with
    -fno-slp-vectorize (csrc/Makefile), and the only packed fp32 instructions left are in the kernels that use
    two-element vectors by hand and were stress-tested under sharing."""
    import re
    makefile = open(os.path.join(ROOT, "superresolution_amd", "csrc", "Makefile")).read()
    flagged = [ln for ln in makefile.splitlines() if ln.rstrip().endswith("CXXFLAGS += -fno-slp-vectorize")]
    assert len(flagged) == 1
    for obj in ("memops.o", "loss.o", "conv_direct.o", "conv3x3_mfma.o", "conv3x3_pair.o", "conv_gemm.o", "conv1x1_chain.o", "conv3x3_wgrad.o"):
        assert obj in flagged[0], obj
    allowed = ("upsample_int_nhwc_kernel", "layernorm_cf", "ssim_fwd_kernel", "ssim_bwd_kernel", "ssim_fwd_stream_kernel", "ssim_bwd_stream_kernel")

    def packed_by_kernel(name, extra):
        out = tmp_path / (name + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", *extra, "-S",
                        "--cuda-device-only", os.path.join(ROOT, "superresolution_amd", "csrc", name), "-o", str(out)],
                       check=True, capture_output=True, timeout=900)
        counts, cur = {}, None
        for ln in open(out):
            m = re.match(r"^([_A-Za-z0-9]+):", ln)
            if m:
                cur = m.group(1)
            elif re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln) and cur:
                counts[cur] = counts.get(cur, 0) + 1
        return counts

    with ThreadPoolExecutor(2) as ex:
        res = list(ex.map(lambda a: packed_by_kernel(*a), [("memops.hip", ["-ffp-contract=off"]), ("loss.hip", []), ("conv_direct.hip", [])]))
    for counts in res:
        for kernel, n in counts.items():
            assert any(a in kernel for a in allowed), (kernel, n)
