"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the module tree /
state_dict / RNG-order contract holds, and the product path refuses to run without a GPU."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_npz
from oracle import unet_oracle as O


def test_library_exports_every_declared_symbol():
    from resdepth_amd import _lib
    lib = _lib.load()            # raises if the .so is missing or a symbol is absent
    hdr = open(os.path.join(ROOT, "include", "resdepth_hip.h")).read()
    declared = set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rd_version() >= 100


@pytest.mark.parametrize("c,sk,d,bias", [(3, 64, 5, True), (1, 8, 3, False), (2, 16, 4, True)])
def test_module_tree_matches_reference_layout(c, sk, d, bias):
    from resdepth_amd import UNet
    spec = O.Spec(n_input_channels=c, start_kernel=sk, depth=d, bias_conv_layer=bias)
    torch.manual_seed(3)
    m = UNet(n_input_channels=c, start_kernel=sk, depth=d, bias_conv_layer=bias)
    sd = m.state_dict()
    layout = O.param_layout(spec)
    assert list(sd.keys()) == [k for k, _, _ in layout]
    for k, shape, _ in layout:
        assert tuple(sd[k].shape) == tuple(shape), k
    ref = O.init_state_dict(spec, 3)              # same seed -> identical weights (RNG draw order)
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
    assert [n for n, _ in m.named_parameters()] == O.param_keys(spec)


def test_golden_state_dict_loads():
    import json
    from resdepth_amd import UNet
    g = load_npz("g1_tiny3.npz")
    kwargs = json.loads(str(g["kwargs_json"]))
    m = UNet(**kwargs)
    sd = {k[len("init/"):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("init/")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_constructor_errors_match_reference():
    from resdepth_amd import UNet
    with pytest.raises(ValueError, match="not a valid activation function"):
        UNet(act_fn_encoder="gelu")
    with pytest.raises(ValueError, match="not a valid mode for upsampling"):
        UNet(up_mode="nearest")


def test_no_cpu_fallback():
    from resdepth_amd import UNet, masked_l1_loss, FusedAdam
    m = UNet(n_input_channels=1, start_kernel=4, depth=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_l1_loss(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4), torch.ones(1, 1, 4, 4, dtype=torch.bool),
                       torch.zeros(1), torch.ones(1))
    opt = FusedAdam(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "resdepth_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_shipped_code_objects_hold_no_packed_f32_valu_and_no_spills():
    """csrc/build.sh builds every translation unit with -fno-slp-vectorize: r02 traced a data corruption of the
    two-stream backward to `v_pk_fma_f32 ... op_sel` fed by ds_read2_b32 under co-execution (no root cause), so the
    instruction must not come back through a compiler / flag change in ANY kernel.  scripts/check_isa.sh disassembles the
    built library (llvm-objdump, no GPU needed)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from resdepth_amd import _lib
    r = subprocess.run(["bash", os.path.join(root, "scripts", "check_isa.sh"), _lib.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "packed-f32 VALU 0, scratch 0" in r.stdout


def test_c_abi_consumer_without_torch_compiles_and_links(tmp_path):
    """tests/cabi/consumer.cpp: HIP runtime + include/resdepth_hip.h only (no PyTorch) builds against the in-tree library --
    the header is self-contained and every entry point the consumer uses resolves.  (It runs in tests/test_cabi_consumer_gpu.py.)"""
    import subprocess
    from cabi_build import build_consumer
    exe = build_consumer(tmp_path)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libresdepth_hip.so" in needed and "torch" not in needed and "python" not in needed.lower()



def _run_bench(args, env=None, timeout=240):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)                      # the driver's plain `python3 bench.py --gpus N`: no launcher environment
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                          timeout=timeout)


def test_bench_gpus_n_starts_its_own_ranks_without_a_launcher():
    """`python3 bench.py --gpus N` (the driver's command form) must not need torch.distributed.run.  Here: two
    ranks on CPU (gloo stands in for RCCL in the GPU-less container) rendezvous, all-reduce, and rank 0 prints ONE line."""
    import json
    r = _run_bench(["--gpus", "2", "--rendezvous-only", "--backend", "gloo"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["rendezvous"] == "ok" and rec["world_size_reported"] == 2 and rec["rank_sum"] == 1.0
    assert rec["self_launched"] is True
    assert rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"       # the RCCL precondition of these hosts travels to the ranks


def test_bench_self_launch_propagates_a_failing_rank():
    r = _run_bench(["--gpus", "2", "--rendezvous-only", "--backend", "gloo"], env={"RD_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "rank 1 exited with 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_under_a_launcher_of_the_wrong_size_is_refused():
    r = _run_bench(["--gpus", "4", "--rendezvous-only", "--backend", "gloo"],
                   env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29655"})
    assert r.returncode == 2 and "--nproc-per-node 4" in r.stderr
