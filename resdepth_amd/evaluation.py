"""Masked residual statistics on the GPU: the reference's `compute_residuals` + `get_statistics`
(lib/evaluation.py:11-131) for rasters resident in HBM (a city-scale DSM is 10^8 pixels; the reference does this with
numpy masked arrays and three full sorts on the CPU).  Returns the reference's statistic names as python floats."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, load, ptr, stream_ptr, workspace

_KEYS = ["count_total", "diff_max", "diff_min", "MAE", "RMSE", "absolute_median", "median", "NMAD"]


def _stats(raster, gt, mask, nodata, thr, dev):
    n = raster.numel()
    out = torch.empty(8, dtype=torch.float64, device=dev)
    ws = workspace(load().rd_residual_stats_ws_bytes(n), dev, slot=2)
    check(load().rd_residual_stats(ptr(raster), ptr(gt), ptr(mask), n, float(nodata), float(thr if thr else -1.0), ptr(out),
                                   ws.data_ptr(), ws.numel(), stream_ptr()), "residual_stats")
    return out


def get_statistics(raster, raster_gt, nodata, mask_gt=None, residual_threshold=None, device="cuda"):
    """raster: refined DSM (any float dtype, e.g. the float64 output of predict_linear_blend), raster_gt: reference DSM,
    mask_gt: optional boolean validity mask.  -> dict as lib/evaluation.py:get_statistics (incl. 'truncated' sub-dict
    when residual_threshold is given)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("resdepth_amd.evaluation.get_statistics needs a HIP device (no CPU fallback)")
    r = torch.as_tensor(np.asarray(raster) if not torch.is_tensor(raster) else raster).to(dev, torch.float64).contiguous()
    g = torch.as_tensor(np.asarray(raster_gt) if not torch.is_tensor(raster_gt) else raster_gt).to(dev, torch.float32).contiguous()
    m = None
    if mask_gt is not None:
        m = torch.as_tensor(np.asarray(mask_gt) if not torch.is_tensor(mask_gt) else mask_gt).to(dev).to(torch.uint8).contiguous()
    with torch.cuda.device(r.device):              # raw launches go to the current device's stream
        full = _stats(r, g, m, nodata, None, r.device)
        trunc = _stats(r, g, m, nodata, residual_threshold, r.device) if residual_threshold else None
    vals = full.cpu().tolist()
    stats = {"truncation": bool(residual_threshold)}
    stats.update(dict(zip(_KEYS, vals)))
    if trunc is not None:
        tv = trunc.cpu().tolist()
        stats["truncated"] = {"count_total": tv[0], "threshold": residual_threshold, "MAE": tv[3], "RMSE": tv[4],
                              "absolute_median": tv[5], "median": tv[6], "NMAD": tv[7]}
    return stats
