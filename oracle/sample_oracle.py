"""numpy restatement of the reference's training-sample assembly (TEST INFRASTRUCTURE ONLY):
DsmOrthoDataset.__getitem__ for sampling_strategy='train' (lib/DsmOrthoDataset.py:161-291), _get_dsm_loss_mask
(lib/DsmOrthoDataset.py:434-470), get_transform = ToTensor + Normalize (lib/data_normalization.py:6-26) and the
augmentation chain Rotate -> RandomVerticalFlip -> RandomHorizontalFlip (lib/torch_transforms.py, applied to
cat(mask, target, inputs), lib/DsmOrthoDataset.py:262-273).  Pinned by tests/golden/g9_samples.npz, which was produced by
the reference's own __getitem__ (tests/golden/make_golden_samples.py)."""
from __future__ import annotations

import numpy as np


def augment(stack: np.ndarray, k: int, flip_v: bool, flip_h: bool) -> np.ndarray:
    """[C,T,T]: np.rot90(., k) counter-clockwise, then np.flipud, then np.fliplr, channel by channel."""
    out = np.stack([np.rot90(c, k) for c in stack])
    if flip_v:
        out = out[:, ::-1, :]
    if flip_h:
        out = out[:, :, ::-1]
    return np.ascontiguousarray(out)


def assemble(dsm_in, dsm_gt, orthos_hwv, pos, pair, tile, nodata, dsm_std, ortho_mean, ortho_std, aug=None):
    y, x = pos
    patch = dsm_in[y:y + tile, x:x + tile]
    tgt = dsm_gt[y:y + tile, x:x + tile]
    mask = np.logical_and(tgt != 0, tgt != nodata)                                   # :434-470 (valid = copy of dsm)
    mean = np.ma.mean(np.ma.masked_where(patch == nodata, patch))                    # :193-195, float32
    mean32, std32 = np.float32(mean), np.float32(dsm_std)
    din = ((patch - mean32) / std32).astype(np.float32)
    tg = ((tgt - mean32) / std32).astype(np.float32)
    o = orthos_hwv[y:y + tile, x:x + tile, pair].transpose((2, 0, 1)).astype(np.float32)
    om = np.float32(o.mean() if ortho_mean is None else ortho_mean)                  # :231-236
    o = ((o - om) / np.float32(ortho_std)).astype(np.float32)
    inputs = np.concatenate([din[None], o], 0)
    m = mask[None].astype(np.float32)
    if aug is not None:
        k, fv, fh = aug
        st = augment(np.concatenate([m, tg[None], inputs], 0), int(k), bool(fv), bool(fh))
        m, tgc, inputs = st[0:1], st[1:2], st[2:]
    else:
        tgc = tg[None]
    return {"input": inputs, "target": tgc, "loss_mask": m.astype(bool), "dsm_mean": float(mean)}
