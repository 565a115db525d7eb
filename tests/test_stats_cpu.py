"""Evaluation statistics, CPU side: oracle vs fixtures from the reference's compute_residuals / get_statistics."""
import numpy as np

from conftest import load_npz
from oracle import stats_oracle as E

KEYS = ["count_total", "diff_max", "diff_min", "MAE", "RMSE", "absolute_median", "median", "NMAD"]
TKEYS = ["count_total", "MAE", "RMSE", "absolute_median", "median", "NMAD"]


def test_oracle_statistics_match_reference():
    g = load_npz("g10_stats.npz")
    for i in range(int(g["n"])):
        mask = g.get(f"c{i}/mask")
        r, valid = E.residuals(g[f"c{i}/raster"], g[f"c{i}/gt"], -9999.0, mask)
        st = E.statistics(r, valid)
        np.testing.assert_allclose([st[k] for k in KEYS], g[f"c{i}/stats"], rtol=1e-12, atol=1e-12)
        thr = float(g[f"c{i}/thr"])
        if thr > 0:
            tt = E.statistics(r, valid, thr)
            np.testing.assert_allclose([tt[k] for k in TKEYS], g[f"c{i}/tstats"], rtol=1e-12, atol=1e-12)
