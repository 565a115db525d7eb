"""Tiled inference, CPU side: the oracle restatement and the product's host-side grid generator
against fixtures produced by the reference's own create_regular_grid / _get_blend_weights / predict_linear_blend."""
import numpy as np

from conftest import load_npz
from oracle import blend_oracle as B


def test_regular_grid_matches_reference():
    from resdepth_amd import tiling
    g = load_npz("g6_blend.npz")
    for i in range(int(g["grid_n"])):
        x0, x1, y0, y1, t, s = [int(v) for v in g[f"grid{i}/args"]]
        for fn in (B.regular_grid, tiling.regular_grid):
            pos, reg = fn([(x0, x1)], [(y0, y1)], t, s)
            assert np.array_equal(np.array(pos), g[f"grid{i}/pos"]), (i, fn.__module__)
            assert np.array_equal(np.array(reg), g[f"grid{i}/reg"]), (i, fn.__module__)
    for fn in (B.regular_grid, tiling.regular_grid):
        pos, reg = fn([(0, 40), (50, 90)], [(0, 30), (10, 45)], 16, 8)
        assert np.array_equal(np.array(pos), g["grid_multi/pos"]) and np.array_equal(np.array(reg), g["grid_multi/reg"])


def test_blend_weights_match_reference():
    g = load_npz("g6_blend.npz")
    for i, (t, s, ulx, uly, lrx, lry) in enumerate(g["w_args"].tolist()):
        w = B.blend_weights(t, s, ulx, uly, lrx, lry)
        np.testing.assert_allclose(w, g[f"w{i}"], rtol=0, atol=1e-15, err_msg=str((t, s, ulx, uly, lrx, lry)))


def test_accumulate_matches_reference_predict_linear_blend():
    g = load_npz("g6_blend.npz")
    rows, cols, t, s = [int(v) for v in g["blend/shape"]]
    x = g["blend/tiles"]
    pred = x[:, 0:1] * 0.5 + 0.25 * x[:, 1:2] * x[:, 1:2]          # the fixture's stand-in "model"
    raster = B.accumulate(np.zeros((rows, cols)), pred, g["blend/means"], g["blend/stds"],
                          [tuple(p) for p in g["blend/pos"]], [tuple(r) for r in g["blend/reg"]], t, s)
    np.testing.assert_allclose(raster, g["blend/raster"], rtol=1e-13, atol=1e-10)


def test_weights_are_a_partition_of_unity():
    """Size-independent property: over any grid the blend weights of all tiles sum to exactly 1 per pixel."""
    for (rows, cols, t, s) in [(41, 53, 16, 8), (300, 517, 64, 32), (256, 256, 256, 128), (1000, 700, 256, 128)]:
        pos, reg = B.regular_grid([(0, cols - 1)], [(0, rows - 1)], t, s)
        acc = np.zeros((rows, cols))
        for (y, x), (uly, ulx, lry, lrx) in zip(pos, reg):
            acc[y:y + t, x:x + t] += B.blend_weights(t, s, ulx, uly, lrx, lry)
        np.testing.assert_allclose(acc, 1.0, rtol=0, atol=1e-12)


def test_world8_band_sweep_assembles_the_unsharded_raster():
    """cfg-G's multi-rank sweep at WORLD 8 (tiling.band_shards + the exchange of resdepth_amd/inference.py:_exchange_overlaps),
    executed rank by rank in one process with the oracle's accumulate: every rank blends its tiles into its private band
    [lo, hi), rows of its extent that another rank owns travel to their owner and are added there in ascending sender order,
    every rank delivers the rows it owns -- the assembled raster equals the unsharded sweep (same additions per pixel up to
    their order: 1e-12), on a two-area raster whose bands share rows with one and with two neighbours."""
    from resdepth_amd.tiling import band_shards, regular_grid
    rows, cols, t, s = 1400, 300, 64, 32
    areas_x, areas_y = [(0, 199), (80, 299)], [(0, 799), (600, 1399)]
    pos, reg = regular_grid(areas_x, areas_y, t, s)
    rng = np.random.default_rng(5)
    pred = rng.standard_normal((len(pos), 1, t, t))
    means, stds = rng.standard_normal(len(pos)) * 10 + 400, np.full(len(pos), 3.0)
    full = B.accumulate(np.zeros((rows, cols)), pred, means, stds, pos, reg, t, s)
    for world in (8, 5):
        plan = band_shards(pos, t, rows, world)
        assert plan[0]["monotonic"] and plan[0]["c0"] == 0 and plan[-1]["c1"] == rows
        private = []
        for r, me in enumerate(plan):                       # every rank's sweep of its own tiles into its band-sized raster
            band = np.zeros((max(me["hi"] - me["lo"], 0), cols))
            i0, i1 = me["i0"], me["i1"]
            shifted = [(y - me["lo"], x) for (y, x) in pos[i0:i1]]
            private.append(B.accumulate(band, pred[i0:i1], means[i0:i1], stds[i0:i1], shifted, reg[i0:i1], t, s) if i1 > i0 else band)
        out = np.full((rows, cols), np.nan)
        for r, me in enumerate(plan):
            own = private[r][me["c0"] - me["lo"]:me["c1"] - me["lo"]].copy()
            for q, o in enumerate(plan):                    # ascending sender order, as the exchange adds them
                if q == r or o["y0"] is None:
                    continue
                a, b = max(o["y0"], me["c0"]), min(o["y1"], me["c1"])
                if b > a:
                    own[a - me["c0"]:b - me["c0"]] += private[q][a - o["lo"]:b - o["lo"]]
            out[me["c0"]:me["c1"]] = own
        assert not np.isnan(out).any()
        np.testing.assert_allclose(out, full, rtol=1e-12, atol=1e-9)
