"""Data-parallel training step end to end on the GPU: two ranks (sharing the one GPU of the test box, gloo
transport with CUDA tensors -- the production launch uses RCCL through the same code) run the HIP path with
all-gathered global negatives and SUM-all-reduced gradients; loss and every parameter gradient must equal the
single-process step at the doubled batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_single_process_global_batch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_rank_synchronised_batchnorm_equals_single_process_global_batch():
    """ConvMixer tower (three BatchNorms per layer) under data parallel with `enable_sync_batchnorm`: loss, gradients
    and running statistics of two ranks equal the single-process step at the doubled batch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py"), "--batchnorm"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
