"""CPU oracle for the ResDepth U-Net hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy) of the arithmetic the
reference performs on its hot path (lib/UNet.py forward, the masked L1 loss of
lib/Trainer.py:87-100, torch.optim.Adam as configured by lib/utils.py:329-331).
It exists so the HIP kernels can be checked against the reference's numerics on
a GPU box where /root/reference does not exist.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  Nothing under `resdepth_amd/` imports it; the product path fails
loudly when the HIP library is missing.

Parity pinning: the reference has no tests of its own, so
the oracle is pinned against the reference *itself*: `tests/golden/make_golden.py`
imports `lib/UNet.py` from /root/reference in the build container and writes the
fixtures under `tests/golden/`; `tests/test_oracle_golden.py` asserts this
restatement reproduces them.
"""
