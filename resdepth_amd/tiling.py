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


def band_shards(pos: Sequence[Tuple[int, int]], tile_size: int, rows: int, world: int):
    """Shard a sweep's tile list (in sweep order, `regular_grid`) over `world` ranks by ROW BANDS: contiguous
    chunks cut at tile-row boundaries, balanced by tile count.  A rank's tiles then touch only the raster rows
    [y0, y1) of its band (+ the T - stride rows it shares with each neighbour), so its private raster is band-sized, the
    exchange is the shared rows only, and every rank reads ITS rows back to the host.

    -> list of `world` dicts: tiles [i0, i1) of `pos`; extent [y0, y1) = raster rows the rank's tiles touch; owned [c0, c1) =
    the rows the rank delivers (the cuts c partition [0, rows): c_r = y0 of rank r's first tile, c_0 = 0, c_world = rows);
    lo / hi = the rows its private raster covers = [min(c0, y0), max(c1, y1)).  Ranks beyond the number of tile rows get an
    empty range.  `monotonic` (same on every entry) says whether the cuts are non-decreasing -- false for tile lists whose
    areas go back up the raster, which the caller then sweeps with full-size rasters + a reduce instead."""
    n = len(pos)
    starts = [i for i in range(n) if i == 0 or pos[i][0] != pos[i - 1][0]] + [n]       # first tile of every tile row
    groups = len(starts) - 1
    cuts_i = [0]
    for r in range(1, world):
        # boundary (a tile-row start) whose tile count is closest to the r-th equal share, at least one tile row per rank
        target = n * r / world
        lo_g = min(groups, r)                               # rank r-1 keeps at least one row
        cand = range(max(lo_g, 0), groups + 1)
        best = min(cand, key=lambda g: (abs(starts[g] - target), g)) if groups else 0
        best = max(best, starts.index(cuts_i[-1]) + (1 if cuts_i[-1] < n else 0)) if groups else 0
        cuts_i.append(starts[min(best, groups)])
    cuts_i.append(n)
    out = []
    for r in range(world):
        i0, i1 = cuts_i[r], cuts_i[r + 1]
        ys = [pos[i][0] for i in range(i0, i1)]
        out.append({"i0": i0, "i1": i1, "y0": min(ys) if ys else None, "y1": (max(ys) + tile_size) if ys else None})
    # ownership cuts: rank r starts owning at its first tile's row; empty ranks own nothing (cut = the next non-empty one's)
    c = [0] * (world + 1)
    c[world] = rows
    for r in range(world - 1, 0, -1):
        c[r] = out[r]["y0"] if out[r]["y0"] is not None else c[r + 1]
    mono = all(c[r] <= c[r + 1] for r in range(world))
    for r in range(world):
        e = out[r]
        e["c0"], e["c1"] = c[r], c[r + 1]
        e["lo"] = min(e["c0"], e["y0"]) if e["y0"] is not None else e["c0"]
        e["hi"] = max(e["c1"], e["y1"]) if e["y1"] is not None else e["c1"]
        e["monotonic"] = mono
    return out
