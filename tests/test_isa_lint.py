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
KERNELS = ["conv3x3_mfma.hip", "conv3x3_wgrad.hip", "conv3x3_pair.hip", "conv_gemm.hip"]


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
