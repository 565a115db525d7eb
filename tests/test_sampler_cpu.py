"""Sample assembly, CPU side: the oracle against samples produced by the reference's own
DsmOrthoDataset.__getitem__ + torch_transforms (tests/golden/g9_samples.npz)."""
import numpy as np

from conftest import load_npz
from oracle import sample_oracle as S


def test_oracle_reproduces_reference_samples():
    g = load_npz("g9_samples.npz")
    t = int(g["tile"])
    for i, (pos, pair, aug) in enumerate(zip(g["pos"], g["pairs"], g["aug"])):
        s = S.assemble(g["dsm_in"], g["dsm_gt"], g["orthos"], tuple(pos), list(pair), t, g["nodata"], g["dsm_std"],
                       g["ortho_mean"], g["ortho_std"], tuple(aug))
        np.testing.assert_array_equal(s["loss_mask"], g[f"s{i}/loss_mask"])
        np.testing.assert_allclose(s["input"], g[f"s{i}/input"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(s["target"], g[f"s{i}/target"], rtol=0, atol=1e-6)
        assert abs(s["dsm_mean"] - float(g[f"s{i}/dsm_mean"])) <= 1e-6 * abs(s["dsm_mean"])
    for i in range(3):                                  # no augmentation, per-patch ortho mean
        s = S.assemble(g["dsm_in"], g["dsm_gt"], g["orthos"], tuple(g["pos"][i]), list(g["pairs"][i]), t, g["nodata"],
                       g["dsm_std"], None, g["ortho_std"], None)
        np.testing.assert_array_equal(s["loss_mask"], g[f"n{i}/loss_mask"])
        np.testing.assert_allclose(s["input"], g[f"n{i}/input"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(s["target"], g[f"n{i}/target"], rtol=0, atol=1e-6)
