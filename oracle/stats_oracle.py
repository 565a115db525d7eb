"""numpy restatement of the reference's residual statistics (TEST INFRASTRUCTURE ONLY): compute_residuals and
get_statistics (lib/evaluation.py:11-131).  Pinned by tests/golden/g10_stats.npz (produced by the reference functions).

Note the reference's NMAD is 1.4826 * median(|r - absolute_median|) (it subtracts the median ABSOLUTE error,
lib/evaluation.py:112-113), reproduced as is."""
from __future__ import annotations

import numpy as np


def residuals(raster, raster_gt, nodata, mask_gt=None):
    """-> (r float64 flat, valid bool flat): valid where neither raster holds nodata and (if given) mask_gt is True."""
    valid = (raster != nodata) & (raster_gt != nodata)
    if mask_gt is not None:
        valid &= mask_gt.astype(bool)
    return (raster.astype(np.float64) - raster_gt.astype(np.float64)).ravel(), valid.ravel()


def _median(v):
    v = np.sort(v)
    n = v.size
    return float("nan") if n == 0 else 0.5 * (v[(n - 1) // 2] + v[n // 2])


def statistics(r, valid, threshold=None):
    """-> dict with count_total, diff_max, diff_min, MAE, RMSE, absolute_median, median, NMAD of the valid residuals;
    with a threshold the residuals outside [-threshold, threshold] are dropped first (truncate_residuals)."""
    v = r[valid]
    if threshold is not None:
        v = v[np.abs(v) <= threshold]
    a = np.abs(v)
    am = _median(a)
    return {"count_total": float(v.size), "diff_max": float(v.max()), "diff_min": float(v.min()), "MAE": float(a.mean()),
            "RMSE": float(np.sqrt((a * a).mean())), "absolute_median": am, "median": _median(v),
            "NMAD": 1.4826 * _median(np.abs(v - am))}
