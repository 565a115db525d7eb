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
