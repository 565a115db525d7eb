#!/usr/bin/env python3
"""Golden fixtures for the sample-assembly row: training samples produced by the REFERENCE's own
`DsmOrthoDataset.__getitem__` (lib/DsmOrthoDataset.py:161-291) and `lib/torch_transforms.py` (build container only).

The dataset class cannot be constructed here (its constructor reads GeoTIFFs through GDAL), so an instance is created
with object.__new__ and the attributes its constructor would set are filled with synthetic in-memory rasters; then the
unmodified `__getitem__` runs.  GDAL / easydict / torchsummary / tensorboard are absent: empty stand-in modules are
registered as in make_golden_blend.py.  torchvision is absent too, and `__getitem__` really calls
`transforms.Compose([ToTensor(), Normalize(mean, std)])` (lib/data_normalization.py:6-26), so three FUNCTIONAL stand-ins
are provided with torchvision's documented semantics for a float32 HxW array: ToTensor -> tensor[1,H,W] (no scaling for
float input), Normalize -> (t - mean) / std per channel in float32, Compose -> apply in order.
Output: g9_samples.npz (rasters, positions, augmentation draws and the produced sample dicts; data only)."""
import os
import random
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
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


class _Compose:
    def __init__(self, ts):
        self.transforms = ts

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, a):
        a = np.asarray(a)
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.unsqueeze(0) if t.dim() == 2 else t.permute(2, 0, 1).contiguous()


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)


stub("easydict", EasyDict=_EasyDict)
stub("osgeo", gdal=stub("osgeo.gdal", GA_ReadOnly=0))
stub("torchvision", transforms=stub("torchvision.transforms", Compose=_Compose, ToTensor=_ToTensor, Normalize=_Normalize))
stub("torchsummary", summary=lambda *a, **k: None)
import torch.utils  # noqa: E402
torch.utils.tensorboard = stub("torch.utils.tensorboard", SummaryWriter=type("SummaryWriter", (), {}))

from lib.DsmOrthoDataset import DsmOrthoDataset  # noqa: E402  (reference)

T, H, W, V = 16, 48, 64, 3
rng = np.random.RandomState(0)
dsm_in = (rng.randn(H, W) * 4 + 430).astype(np.float32)
dsm_gt = (dsm_in + rng.randn(H, W) * 1.5).astype(np.float32)
NODATA = np.float32(-9999.0)
dsm_in[5:8, 10:14] = NODATA
dsm_gt[20:23, 30:33] = NODATA
dsm_gt[40, 50] = 0.0                      # the reference's mask also drops exact zeros (valid = copy of dsm; mask1 = valid != 0)
orthos = (rng.rand(H, W, V) * 200 + 20).astype(np.float32)

ds = object.__new__(DsmOrthoDataset)
ds.tile_size, ds.sampling_strategy, ds.augment = T, "train", True
ds.input_channels = "geom-stereo"
ds.transform_dsm, ds.transform_orthos = True, True
ds.dsm_mean, ds.dsm_std = None, np.asarray(3.25).astype(np.float32)
ds.ortho_mean, ds.ortho_std = np.asarray(118.5).astype(np.float32), np.asarray(41.0).astype(np.float32)
ds.permute_images_within_pair = False
ds.raster_gt = "in-memory"
ds.dsm_input, ds.dsm_target, ds.orthos, ds.nodata = dsm_in, dsm_gt, orthos, np.array(NODATA)
ds.image_pairs = [[0, 1], [1, 2], [0, 2]]
positions = [(0, 0), (3, 7), (32, 48), (17, 40), (30, 20), (0, 48), (12, 3), (25, 25)]
ds.patch_position = positions
ds.image_pair_indices = np.array([0, 1, 2, 0, 1, 2, 0, 1], dtype=np.int64)

out = {"dsm_in": dsm_in, "dsm_gt": dsm_gt, "orthos": orthos, "nodata": NODATA, "tile": np.array(T),
       "dsm_std": np.float32(3.25), "ortho_mean": np.float32(118.5), "ortho_std": np.float32(41.0),
       "pos": np.array(positions), "pairs": np.array([ds.image_pairs[i] for i in ds.image_pair_indices])}
aug = []
for i in range(len(positions)):
    random.seed(1000 + i)
    # the draws __getitem__ will make: Rotate() -> random.randint(0,3) at construction; then the two flips'
    # random.random() < 0.5 at call time (vertical first, then horizontal)
    k = random.randint(0, 3)
    fv = random.random() < 0.5
    fh = random.random() < 0.5
    aug.append((k, int(fv), int(fh)))
    random.seed(1000 + i)
    smp = ds[i]
    out[f"s{i}/input"] = smp["input"].numpy()
    out[f"s{i}/target"] = smp["target"].numpy()
    out[f"s{i}/loss_mask"] = smp["loss_mask"].numpy()
    out[f"s{i}/dsm_mean"] = np.float64(smp["dsm_mean"])
out["aug"] = np.array(aug)
# a second pass without augmentation and with per-patch ortho mean
ds.augment, ds.ortho_mean = False, None
for i in range(3):
    smp = ds[i]
    out[f"n{i}/input"] = smp["input"].numpy()
    out[f"n{i}/target"] = smp["target"].numpy()
    out[f"n{i}/loss_mask"] = smp["loss_mask"].numpy()
np.savez_compressed(os.path.join(HERE, "g9_samples.npz"), **out)
print("g9_samples.npz", os.path.getsize(os.path.join(HERE, "g9_samples.npz")), "bytes; aug draws", aug)
