"""The ctypes example of INTEGRATION.md, executed verbatim in a fresh interpreter (nothing of this package imported:
the reader's situation) and checked against the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.test_abi_cpu import ROOT, integration_snippet

pytestmark = pytest.mark.gpu

_DRIVER = r"""
import sys, numpy as np, torch
z = np.load(sys.argv[1])
x = torch.from_numpy(z["x"]).cuda().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)     # (N,H,W,64) bf16
conv = torch.nn.Conv2d(64, 64, 3, 1, 1).cuda()
with torch.no_grad():
    conv.weight.copy_(torch.from_numpy(z["w"])); conv.bias.copy_(torch.from_numpy(z["b"]))
exec(open(sys.argv[2]).read())
torch.cuda.synchronize()
assert "superresolution_amd" not in sys.modules
np.save(sys.argv[3], y.float().permute(0, 3, 1, 2).cpu().numpy())
"""


def test_integration_example_runs_as_written_and_matches_the_oracle(tmp_path):
    rng = np.random.default_rng(11)
    bf = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()
    x = bf(rng.uniform(-1, 1, (2, 64, 21, 150)))
    w = bf(rng.standard_normal((64, 64, 3, 3)) * 0.06)
    b = bf(rng.standard_normal(64) * 0.1)
    np.savez(tmp_path / "in.npz", x=x, w=w, b=b)
    (tmp_path / "snippet.py").write_text(integration_snippet())
    (tmp_path / "driver.py").write_text(_DRIVER)
    r = subprocess.run([sys.executable, str(tmp_path / "driver.py"), str(tmp_path / "in.npz"), str(tmp_path / "snippet.py"),
                        str(tmp_path / "y.npy")], cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    y = np.load(tmp_path / "y.npy")
    ref = orc.conv2d(x, w, b, 1, 1, act=1)
    assert y.shape == ref.shape
    assert float(np.abs(y - ref).max()) <= 2e-2      # bf16 output
