#!/usr/bin/env python3
"""Golden fixtures for the evaluation-statistics row: produced by the REFERENCE's own
`compute_residuals` / `get_statistics` (lib/evaluation.py:11-131), build container only.  Import stand-ins for the
absent GDAL / easydict / torchvision / torchsummary / tensorboard modules as in make_golden_blend.py (none of them is
touched by the two functions, except EasyDict which is only the result container).  Output: g10_stats.npz (data only)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    __setattr__ = dict.__setitem__


stub("easydict", EasyDict=_EasyDict)
stub("osgeo", gdal=stub("osgeo.gdal", GA_ReadOnly=0))
stub("torchvision", transforms=stub("torchvision.transforms", Compose=type("Compose", (), {}),
                                    ToTensor=type("ToTensor", (), {}), Normalize=type("Normalize", (), {})))
stub("torchsummary", summary=lambda *a, **k: None)
import torch.utils  # noqa: E402
torch.utils.tensorboard = stub("torch.utils.tensorboard", SummaryWriter=type("SummaryWriter", (), {}))
from lib import evaluation  # noqa: E402  (reference)

KEYS = ["count_total", "diff_max", "diff_min", "MAE", "RMSE", "absolute_median", "median", "NMAD"]
TKEYS = ["count_total", "MAE", "RMSE", "absolute_median", "median", "NMAD"]
rng = np.random.RandomState(3)
out = {}
cases = [(37, 53, True, 2.0), (64, 64, False, 1.5), (5, 7, True, None), (120, 200, True, 3.0)]
for i, (h, w, with_mask, thr) in enumerate(cases):
    gt = (rng.randn(h, w) * 5 + 420).astype(np.float32)
    raster = gt.astype(np.float64) + rng.standard_t(3, size=(h, w)) * 0.8
    nodata = -9999.0
    gt[rng.rand(h, w) < 0.03] = nodata
    raster[rng.rand(h, w) < 0.02] = nodata
    mask = (rng.rand(h, w) > 0.1) if with_mask else None
    res = evaluation.compute_residuals(raster, gt, nodata, mask)
    st = evaluation.get_statistics(res, thr)
    out[f"c{i}/raster"], out[f"c{i}/gt"] = raster, gt
    if mask is not None:
        out[f"c{i}/mask"] = mask
    out[f"c{i}/thr"] = np.float64(-1.0 if thr is None else thr)
    out[f"c{i}/stats"] = np.array([float(st[k]) for k in KEYS])
    if thr:
        out[f"c{i}/tstats"] = np.array([float(st.truncated[k]) for k in TKEYS])
out["n"] = np.array(len(cases))
np.savez_compressed(os.path.join(HERE, "g10_stats.npz"), **out)
print("g10_stats.npz", os.path.getsize(os.path.join(HERE, "g10_stats.npz")), [out[f"c{i}/stats"][:4] for i in range(2)])
