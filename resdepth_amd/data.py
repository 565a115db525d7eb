"""Synthetic stand-in for the reference's DsmOrthoDataset (lib/DsmOrthoDataset.py needs GDAL rasters, which are
out of scope): produces samples with the exact dict contract of `DsmOrthoDataset.__getitem__`
(lib/DsmOrthoDataset.py:281-291) so that a torch DataLoader's default collate yields the batch dict the Trainer
consumes (lib/Trainer.py:102-108,174-175)."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


class SyntheticDsmOrthoDataset(Dataset):
    """n_samples random tiles: input [C,T,T] (channel 0 = normalised DSM, then ortho-images), target [1,T,T],
    loss_mask bool [1,T,T] (~5 % nodata), per-sample dsm_mean / dsm_std, patch offsets and valid-pixel box."""

    def __init__(self, n_samples: int, n_input_channels: int = 3, tile_size: int = 256, seed: int = 0,
                 nodata_frac: float = 0.05, dsm_std: float = 3.0):
        self.n, self.c, self.t = int(n_samples), int(n_input_channels), int(tile_size)
        self.seed, self.nodata_frac, self.dsm_std = int(seed), float(nodata_frac), float(dsm_std)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i))
        x = torch.randn(self.c, self.t, self.t, generator=g)
        y = x[0:1] + 0.3 * torch.randn(1, self.t, self.t, generator=g)
        mask = torch.rand(1, self.t, self.t, generator=g) > self.nodata_frac
        return {
            "input": x, "target": y, "loss_mask": mask,
            "dsm_mean": torch.randn((), generator=g, dtype=torch.float64) * 50.0,
            "dsm_std": torch.tensor(self.dsm_std),
            "patch_offset_x": torch.tensor(0), "patch_offset_y": torch.tensor(0),
            "nodata": torch.tensor(-9999.0),
            "patch_valid_pixels_uly": torch.tensor(0), "patch_valid_pixels_ulx": torch.tensor(0),
            "patch_valid_pixels_lry": torch.tensor(self.t), "patch_valid_pixels_lrx": torch.tensor(self.t),
        }


def synthetic_batch(n: int, c: int, t: int, seed: int = 1234, nodata_frac: float = 0.05):
    """One DataLoader-shaped synthetic batch: randn tiles, target = DSM channel + noise, ~5 % nodata in
    the loss mask, per-sample dsm_mean / dsm_std -- the keys DsmOrthoDataset.__getitem__ + default collate produce
    (lib/DsmOrthoDataset.py:281-291).  Used by bench.py's GPU leg (the oracle keeps its own generator for the tests)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, t, t, generator=g)
    y = x[:, 0:1] + 0.3 * torch.randn(n, 1, t, t, generator=g)
    mask = torch.rand(n, 1, t, t, generator=g) > nodata_frac
    return {"input": x, "target": y, "loss_mask": mask,
            "dsm_mean": torch.randn(n, generator=g, dtype=torch.float64) * 50.0, "dsm_std": torch.full((n,), 3.0)}
