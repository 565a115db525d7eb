"""Evaluation statistics on the GPU (-m gpu) against fixtures from the reference's compute_residuals / get_statistics
and, at city scale, against the oracle."""
import numpy as np
import pytest

from conftest import load_npz
from oracle import stats_oracle as E

pytestmark = pytest.mark.gpu
KEYS = ["count_total", "diff_max", "diff_min", "MAE", "RMSE", "absolute_median", "median", "NMAD"]
TKEYS = ["count_total", "MAE", "RMSE", "absolute_median", "median", "NMAD"]


def test_gpu_statistics_match_reference_fixture():
    from resdepth_amd.evaluation import get_statistics
    g = load_npz("g10_stats.npz")
    for i in range(int(g["n"])):
        thr = float(g[f"c{i}/thr"])
        st = get_statistics(g[f"c{i}/raster"], g[f"c{i}/gt"], -9999.0, g.get(f"c{i}/mask"), thr if thr > 0 else None)
        np.testing.assert_allclose([st[k] for k in KEYS], g[f"c{i}/stats"], rtol=1e-12, atol=1e-12)
        if thr > 0:
            np.testing.assert_allclose([st["truncated"][k] for k in TKEYS], g[f"c{i}/tstats"], rtol=1e-12, atol=1e-12)


def test_large_raster_against_oracle_and_edge_cases():
    from resdepth_amd.evaluation import get_statistics
    rng = np.random.RandomState(1)
    h, w = 2048, 3000                                   # 6.1 M pixels, even and odd valid counts via the mask
    gt = (rng.randn(h, w) * 5 + 400).astype(np.float32)
    raster = gt.astype(np.float64) + rng.laplace(size=(h, w)) * 0.7
    raster[rng.rand(h, w) < 0.01] = -9999.0
    mask = rng.rand(h, w) > 0.2
    r, valid = E.residuals(raster, gt, -9999.0, mask)
    ref, reft = E.statistics(r, valid), E.statistics(r, valid, 1.0)
    st = get_statistics(raster, gt, -9999.0, mask, 1.0)
    np.testing.assert_allclose([st[k] for k in KEYS], [ref[k] for k in KEYS], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose([st["truncated"][k] for k in TKEYS], [reft[k] for k in TKEYS], rtol=1e-12, atol=1e-12)
    # medians are exact order statistics (bit-equal), ties and negative zeros included
    assert st["median"] == ref["median"] and st["absolute_median"] == ref["absolute_median"]
    tiny = get_statistics(np.array([[1.0, -2.0, 0.0, -0.0, 5.0]]), np.zeros((1, 5), np.float32), -9999.0)
    assert tiny["count_total"] == 5 and tiny["median"] == 0.0 and tiny["absolute_median"] == 1.0
    empty = get_statistics(np.full((2, 3), -9999.0), np.zeros((2, 3), np.float32), -9999.0)
    assert empty["count_total"] == 0 and np.isnan(empty["median"]) and np.isnan(empty["MAE"])
