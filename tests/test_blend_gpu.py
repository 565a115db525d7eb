"""Tiled inference on the GPU (-m gpu): the accumulate kernel against the oracle / the reference fixture, and the whole
predict_linear_blend sweep (HIP U-Net in eval mode) against the oracle pipeline."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from conftest import load_npz
from oracle import blend_oracle as B
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_accumulate_kernel_against_reference_fixture():
    from resdepth_amd import ops
    g = load_npz("g6_blend.npz")
    rows, cols, t, s = [int(v) for v in g["blend/shape"]]
    x = torch.from_numpy(g["blend/tiles"])
    pred = (x[:, 0:1] * 0.5 + 0.25 * x[:, 1:2] * x[:, 1:2]).contiguous()
    raster = torch.zeros(rows, cols, dtype=torch.float64, device=DEV)
    pos = torch.from_numpy(g["blend/pos"]).to(torch.int32).to(DEV)
    reg = torch.from_numpy(g["blend/reg"]).to(torch.int32).to(DEV)
    mean = torch.from_numpy(g["blend/means"]).float().to(DEV)
    std = torch.from_numpy(g["blend/stds"]).to(DEV)
    for lo in range(0, pred.shape[0], 7):                      # ragged batches, same result
        hi = min(lo + 7, pred.shape[0])
        ops.blend_accumulate(pred[lo:hi].to(DEV), mean[lo:hi].contiguous(), std[lo:hi].contiguous(),
                             pos[lo:hi].contiguous(), reg[lo:hi].contiguous(), t, s, raster)
    np.testing.assert_allclose(raster.cpu().numpy(), g["blend/raster"], rtol=1e-13, atol=1e-10)


def test_constant_prediction_blends_to_constant():
    """Partition of unity at a realistic size (T=256, stride 128, 1000x1300 raster)."""
    from resdepth_amd import ops, tiling
    rows, cols, t, s = 1000, 1300, 256, 128
    pos, reg = tiling.regular_grid([(0, cols - 1)], [(0, rows - 1)], t, s)
    n = len(pos)
    raster = torch.zeros(rows, cols, dtype=torch.float64, device=DEV)
    pred = torch.zeros(n, 1, t, t, device=DEV)
    ops.blend_accumulate(pred, torch.full((n,), 412.5, device=DEV), torch.ones(n, device=DEV),
                         torch.tensor(pos, dtype=torch.int32, device=DEV), torch.tensor(reg, dtype=torch.int32, device=DEV),
                         t, s, raster)
    np.testing.assert_allclose(raster.cpu().numpy(), 412.5, rtol=0, atol=1e-9)


def test_predict_linear_blend_against_oracle_pipeline():
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend
    kw = dict(n_input_channels=2, start_kernel=8, depth=2, bias_conv_layer=True)
    spec = O.Spec(**kw)
    torch.manual_seed(3)
    model = UNet(**kw)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for k in sd:                                             # non-trivial running statistics for eval-mode BN
        if k.endswith("running_mean"):
            sd[k] = torch.linspace(-0.1, 0.1, sd[k].numel())
        if k.endswith("running_var"):
            sd[k] = torch.linspace(0.8, 1.3, sd[k].numel())
    model.load_state_dict(sd)
    ds = SyntheticRasterTiles(150, 203, 2, tile_size=64, seed=5)
    out = predict_linear_blend(DataLoader(ds, batch_size=5, shuffle=False), model)
    assert out.shape == (150, 203) and out.dtype == np.float64
    ref = np.zeros((150, 203))
    for i in range(len(ds)):
        smp = ds[i]
        with torch.no_grad():
            yp = O.forward(dict(sd), smp["input"][None], spec, training=False)
        B.accumulate(ref, yp.numpy(), [float(smp["dsm_mean"])], [float(smp["dsm_std"])], [ds.pos[i]], [ds.reg[i]],
                     64, 32)
    assert np.abs(out - ref).max() <= 1e-4                   # metres; forward noise (~1e-6) x std
    again = predict_linear_blend(DataLoader(ds, batch_size=3, shuffle=False), model)
    assert np.abs(out - again).max() <= 1e-9                 # batch size does not matter


def test_predict_linear_blend_full_architecture_multi_area():
    """BASELINE.json configs[4] (cfg-G) at the REAL architecture: 3-channel depth-5 U-Net on 256x256 tiles at stride 128,
    a raster swept as two areas (the reference grids every allowed area separately, lib/rasterutils.py:100-191) with
    shifted border tiles, ragged batches (5, 5, ..., remainder) and non-trivial running statistics; against the oracle
    forward + the oracle blend, tile by tile."""
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend
    kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
    spec = O.Spec(**kw)
    torch.manual_seed(11)
    model = UNet(**kw)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    for k in [k for k in sd if k.endswith("running_mean")]:  # eval-mode BN with statistics / affine terms that matter
        pre = k[:-len("running_mean")]
        sd[k] = torch.randn(sd[k].shape, generator=g) * 0.2
        sd[pre + "running_var"] = torch.rand(sd[k].shape, generator=g) * 0.8 + 0.4
        sd[pre + "weight"] = torch.rand(sd[k].shape, generator=g) + 0.5
        sd[pre + "bias"] = torch.randn(sd[k].shape, generator=g) * 0.1
    model.load_state_dict(sd)
    rows, cols = 640, 900
    areas = [((0, 559), (0, 383)), ((300, 899), (200, 639))]          # two overlapping areas, neither a tile multiple
    ds = SyntheticRasterTiles(rows, cols, 3, tile_size=256, seed=5, areas=areas)
    assert len(ds) >= 12
    out = predict_linear_blend(DataLoader(ds, batch_size=5, shuffle=False), model)
    ref = np.zeros((rows, cols))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}     # fp64 oracle: no noise of its own
    for i in range(len(ds)):
        smp = ds[i]
        with torch.no_grad():
            yp = O.forward(dict(sd64), smp["input"][None].double(), spec, training=False)
        B.accumulate(ref, yp.numpy(), [float(smp["dsm_mean"])], [float(smp["dsm_std"])], [ds.pos[i]], [ds.reg[i]], 256, 128)
    # north_star: <= 1e-4 residual-height deviation, in metres (forward noise ~1e-5 normalised x std 3; where the two
    # areas overlap the raster holds the sum of two partitions of unity)
    assert np.abs(out - ref).max() <= 1e-4, np.abs(out - ref).max()
    again = predict_linear_blend(DataLoader(ds, batch_size=32, shuffle=False), model)
    assert np.abs(out - again).max() <= 1e-9                 # one full batch == ragged batches (tiles independent in eval)


def test_sweep_reduce_on_rccl_world1_equals_the_plain_sweep():
    """cfg-G's rank-sharded sweep ends in torch.distributed.reduce on RCCL (resdepth_amd/inference.py); with one GPU the
    only RCCL world is 1, where the reduce must be the identity: same raster checksum as the sweep without a process group.
    (World sizes 2 and 4 run on one GPU over gloo: tests/test_dp_world2_gpu.py.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [os.path.join(root, "bench.py"), "--infer", "--gpus", "1", "--raster", "1024", "--steps", "1", "--warmup", "1", "--no-prof"]
    outs = []
    for cmd in ([sys.executable] + common,
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", "29583"] + common + ["--force-dist"]):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[1]["dist"] == {"backend": "nccl", "world_size_reported": 1} and outs[0]["dist"] is None
    assert outs[0]["raster_checksum"] == outs[1]["raster_checksum"]


def test_cfg_g_at_the_survey_size_properties():
    """BASELINE configs[4] at SURVEY 8d's size -- 8192 x 8192 raster, 3 969 tiles of 256 x 256 at stride 128, the cfg-S architecture
    in eval mode -- through size-independent properties (an oracle sweep of 3 969 tiles is hours of CPU):
      * the sweep does not depend on how the tiles are batched (32 per batch vs ragged batches of 13): same raster BITS;
      * tile shards (0, 2) + (1, 2) sum to the unsharded raster (what the rank-sharded sweep reduces on rank 0);
      * with the network's residual branch switched off (last layer zeroed) the prediction is the input DSM, and blending the
        de-normalised tiles must give back the raster itself: partition of unity of the blend weights over the whole grid."""
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend
    R = 8192
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(DEV).eval()

    class Loader(list):
        dataset = None

    def staged(ds, bs):
        ld = Loader({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()} for b in DataLoader(ds, batch_size=bs, shuffle=False))
        ld.dataset = ds
        return ld

    ds = SyntheticRasterTiles(R, R, 3, tile_size=256, seed=1)
    assert len(ds) == 3969
    full = predict_linear_blend(staged(ds, 32), model)
    assert full.shape == (R, R) and np.isfinite(full).all()
    ragged = predict_linear_blend(staged(ds, 13), model)
    assert np.array_equal(full, ragged)
    del ragged
    def shard(r):                                    # every second tile of the SAME dataset (what shard=(r, 2) builds; no second raster)
        import copy
        sh = copy.copy(ds)
        sh.pos, sh.reg = ds.pos[r::2], ds.reg[r::2]
        return sh

    parts = [predict_linear_blend(staged(shard(r), 32), model) for r in range(2)]
    np.testing.assert_allclose(parts[0] + parts[1], full, rtol=1e-9, atol=1e-9)
    del parts
    with torch.no_grad():
        model.last_layer.weight.zero_()
        model.last_layer.bias.zero_()
    ident = predict_linear_blend(staged(ds, 32), model)
    dsm = ds.raster[0].double().numpy()
    assert float(np.abs(ident - dsm).max()) <= 2e-4 * 3.0 + 1e-3          # fp32 normalise / de-normalise of heights around 400 m


def test_host_raster_fresh_by_default_and_reused_on_request():
    """predict_linear_blend returns a FRESH host array per call (the reference's behaviour, lib/evaluation.py:510-513): a caller
    holding the previous result must not see it change; `host="reuse"` (or a HostRaster handed in) delivers into the same
    pinned memory every time, for callers that sweep many rasters of one shape."""
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend
    from resdepth_amd.inference import HostRaster
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True).to(DEV).eval()
    ds = SyntheticRasterTiles(96, 160, 3, tile_size=32, seed=3)
    ld = DataLoader(ds, batch_size=7, shuffle=False)
    a = predict_linear_blend(ld, model)
    keep = a.copy()
    b = predict_linear_blend(ld, model)
    assert not np.shares_memory(a, b) and np.array_equal(a, keep) and np.array_equal(a, b)
    c = predict_linear_blend(ld, model, host="reuse")
    d = predict_linear_blend(ld, model, host="reuse")
    assert np.shares_memory(c, d) and np.array_equal(c, keep)
    h = HostRaster(96, 160, shared=False)
    e = predict_linear_blend(ld, model, host=h)
    assert np.shares_memory(e, h.array) and np.array_equal(e, keep)
    # a HostRaster of another shape is not used (and not written)
    other = HostRaster(64, 64, shared=False)
    other.array[:] = 7.0
    f = predict_linear_blend(ld, model, host=other)
    assert f.shape == (96, 160) and np.array_equal(f, keep) and float(other.array.min()) == 7.0
