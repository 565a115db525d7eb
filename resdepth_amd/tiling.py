"""Regular (overlapping) tile grids for tiled inference -- the index arithmetic of the reference's
`create_regular_grid` (lib/rasterutils.py:100-191): stride T/2 by default for inference
(lib/DsmOrthoDataset.py:99-100), last row / column shifted inwards to end on the region border, plus the
per-tile "region without overlap" box that drives the linear blend weights."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def _axis(lo: int, hi: int, tile: int, stride: int):
    """1-D sweep over the inclusive range [lo, hi] -> [(start, border_lo, border_hi)]."""
    out = []
    start, end = lo, lo
    b_lo, b_hi = 0, stride - 1
    while end < hi:
        end = start + tile - 1
        s, bl, bh = start, b_lo, b_hi
        if end >= hi:                       # shift the last tile inwards
            bl = b_lo + (end - hi)
            end = hi
            s = hi - tile + 1
            bh = tile - 1
        out.append((s, bl, bh))
        start = s + stride if s == start else start + stride
        b_lo = tile - stride
    return out


def regular_grid(x_extent: Sequence[Tuple[int, int]], y_extent: Sequence[Tuple[int, int]], tile_size: int,
                 stride: int | None = None):
    """-> (positions [(uly, ulx)], regions [(border_uly, border_ulx, border_lry, border_lrx)]), row-major per stripe."""
    stride = tile_size if stride is None else stride
    pos: List[Tuple[int, int]] = []
    reg: List[Tuple[int, int, int, int]] = []
    for (x0, x1), (y0, y1) in zip(x_extent, y_extent):
        cols = _axis(int(x0), int(x1), tile_size, stride)
        for (uly, b_uly, b_lry) in _axis(int(y0), int(y1), tile_size, stride):
            for (ulx, b_ulx, b_lrx) in cols:
                pos.append((uly, ulx))
                reg.append((b_uly, b_ulx, b_lry, b_lrx))
    return pos, reg
