"""The exact-f32 MFMA kernels (RD_MFMA=f32) stay a supported arithmetic mode: the library reads the switch once per
process, so the convolution parity tests are re-run in a child process with it set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conv_parity_on_exact_f32_mfma_kernels():
    env = dict(os.environ, RD_MFMA="f32")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ops_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "(conv3x3_fwd_dgrad_wgrad and not 16-64-64) or convt or conv1x1 or known_answers"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_whole_net_training_parity_on_exact_f32_mfma_kernels():
    """the exact-f32 mode (v_mfma_f32_32x32x2_f32: the reference's arithmetic in kind, lib/UNet.py:196-246)
    as a first-class parity configuration -- the twelve reference fixtures (forward, loss, every gradient, pooling indices, BN
    buffers, weights after 1 / 3 optimizer steps), the full-size cfg-S step against the oracle and the reference digest at
    batch 2, and cfg-S at its benchmark batch 32 under the fp64 oracle with imposed decisions, all with RD_MFMA=f32 in a
    child process (the library reads the switch once, at load time)."""
    env = dict(os.environ, RD_MFMA="f32")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_unet_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "tiny_net_against_reference_fixture or full_size_against_oracle_and_reference_digest or "
                              "(other_baseline_configs_against_oracle and cfg-S)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "14 passed" in r.stdout, r.stdout[-500:]
