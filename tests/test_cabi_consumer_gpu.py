"""The C-ABI without PyTorch: tests/cabi/consumer.cpp allocates device memory with the HIP runtime, calls the library through
include/resdepth_hip.h only and checks one convolution layer (forward, data gradient, weight gradient) against loops in double
on the host, then the error contract.  What a C / cgo / JNI host sees."""
import subprocess

import pytest

from cabi_build import build_consumer

pytestmark = pytest.mark.gpu


def test_hip_runtime_only_consumer_runs_one_layer_through_the_c_abi(tmp_path):
    exe = build_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "OK" and sum("max |err|" in ln for ln in lines) == 4 and any("BatchNorm statistics" in ln for ln in lines), r.stdout
    assert any("rd_conv3x3_fwd(cin = 3) ->" in ln and "Cin" in ln for ln in lines), r.stdout
    # rd_set_splitk_workspace: register for (device, stream) / run / un-register, and its argument checks
    assert any("split-K scratch" in ln and "same bits" in ln for ln in lines), r.stdout
    assert any("host pointer -> 1, misaligned -> 1, too small -> 1" in ln for ln in lines), r.stdout
    # host side of the sweep's read-back: page-lock a plain host range, fill it through both copy routes, release it
    assert any("host read-back" in ln and "same bytes" in ln and "null range -> 1, null destination -> 1" in ln for ln in lines), r.stdout
