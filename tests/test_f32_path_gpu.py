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
