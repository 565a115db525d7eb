"""numpy restatement of the reference's tiled inference (TEST INFRASTRUCTURE ONLY):
`create_regular_grid` (lib/rasterutils.py:100-191), `_get_blend_weights` (lib/evaluation.py:516-567) and the
accumulation loop of `predict_linear_blend` (lib/evaluation.py:460-513).  Pinned against fixtures produced by the
reference functions themselves (tests/golden/make_golden_blend.py -> g6_blend.npz, tests/test_blend_cpu.py)."""
from __future__ import annotations

import numpy as np


def regular_grid(x_extent, y_extent, tile_size: int, stride: int | None = None):
    """Regular grid of (overlapping) tiles over inclusive pixel extents; the last tile of each row/column is
    shifted inwards so that it ends on the region border (lib/rasterutils.py:100-191).
    Returns (positions [(uly, ulx)], regions [(b_uly, b_ulx, b_lry, b_lrx)])."""
    stride = tile_size if stride is None else stride
    pos, reg = [], []
    for (x0, x1), (y0, y1) in zip(x_extent, y_extent):
        uly = lry = y0
        b_uly, b_lry = 0, stride - 1
        while lry < y1:
            ulx = lrx = x0
            b_ulx, b_lrx = 0, stride - 1
            lry = uly + tile_size - 1
            if lry >= y1:                      # shift the last row up
                b_uly += lry - y1
                lry = y1
                uly = y1 - tile_size + 1
                b_lry = tile_size - 1
            while lrx < x1:
                lrx = ulx + tile_size - 1
                if lrx >= x1:                  # shift the last column left
                    b_ulx += lrx - x1
                    lrx = x1
                    ulx = x1 - tile_size + 1
                    b_lrx = tile_size - 1
                pos.append((int(uly), int(ulx)))
                reg.append((int(b_uly), int(b_ulx), int(b_lry), int(b_lrx)))
                ulx += stride
                b_ulx = tile_size - stride
            uly += stride
            b_uly = tile_size - stride
    return pos, reg


def _axis_weights(tile_size, overlap, lo, hi):
    """1-D factor of the separable blend weight: ramp up before `lo`, 1 inside [lo, hi], ramp down after `hi`
    (the reference multiplies a left/right and a top/bottom factor into a ones matrix)."""
    w = np.ones(tile_size)
    ramp = np.linspace(0, 1, overlap, endpoint=True)
    if lo > 0:
        if lo == overlap:
            w[0:lo] *= ramp
        else:
            w[lo - overlap:lo] *= ramp
            w[0:lo - overlap] = 0
    if hi < tile_size - 1:
        w[hi + 1:] *= ramp[::-1]
    return w


def blend_weights(tile_size, stride, ulx, uly, lrx, lry):
    """lib/evaluation.py:516-567 -- separable: weights[r, c] = wy[r] * wx[c] (zeros where a shifted border tile
    overlaps more than `overlap` pixels)."""
    overlap = tile_size - stride
    wx = _axis_weights(tile_size, overlap, ulx, lrx)
    wy = _axis_weights(tile_size, overlap, uly, lry)
    return wy[:, None] * wx[None, :]


def accumulate(raster, pred, mean, std, pos, reg, tile_size, stride):
    """raster_out[y:y+T, x:x+T] += (pred*std_i + mean_i)[fp32] * weights[fp64], tile by tile in order
    (lib/evaluation.py:497-511, denormalize_numpy lib/data_normalization.py:41-53)."""
    for i in range(pred.shape[0]):
        y, x = pos[i]
        uly, ulx, lry, lrx = reg[i]
        den = (pred[i, 0].astype(np.float32) * np.float32(std[i]) + np.float32(mean[i])).astype(np.float32)
        raster[y:y + tile_size, x:x + tile_size] += den * blend_weights(tile_size, stride, ulx, uly, lrx, lry)
    return raster
