"""Repeatability of the temporal forward while a second process uses the same GPU (two ranks time-slicing one card is
how the launch path of `bench.py --gpus N` is rehearsed on a one-GPU box).  Starts two processes, so it only runs on
request: SRHIP_SHARING_TEST=1 python -m pytest tests/test_gpu_sharing.py -m gpu."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("SRHIP_SHARING_TEST") != "1", reason="starts two GPU processes; set SRHIP_SHARING_TEST=1")
def test_every_operator_of_the_temporal_forward_repeats_under_gpu_sharing():
    """tools/concurrency_check.py: checksums of every operator output of 12 forwards in each of two concurrent processes;
    any forward that differs from the first names its first differing operator and fails the run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "concurrency_check.py"), "2", "12"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("0 differing") == 2, r.stdout[-2000:]
