"""Training-sample assembly on the GPU: the per-sample work of the reference's
`DsmOrthoDataset.__getitem__` (lib/DsmOrthoDataset.py:161-291) -- patch extraction, masked per-patch mean centring and
division by the global DSM std, ortho-image normalisation, loss mask, rot90 / flip augmentation
(lib/torch_transforms.py) -- as two kernels over rasters that stay resident in HBM (288 GB holds any city raster),
producing the collated batch dict directly on the device.  The reference does this per sample on the CPU with
per-channel numpy loops and cannot feed more than a few hundred tiles/s.

Raster I/O (GeoTIFF via GDAL) and the choice of valid patch positions stay with the caller (out of scope)."""
from __future__ import annotations

import torch

from ._lib import check, load, ptr, stream_ptr


class GpuPatchSampler:
    def __init__(self, dsm_input, dsm_target=None, orthos=None, tile_size: int = 256, nodata: float = -9999.0,
                 dsm_std: float = 1.0, ortho_mean=None, ortho_std: float = 1.0, device="cuda"):
        """dsm_input / dsm_target: [H, W] float32; orthos: [V_total, H, W] float32 (planar).  ortho_mean=None => per-patch
        mean over the views of the sample (lib/DsmOrthoDataset.py:231-233)."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("resdepth_amd.GpuPatchSampler needs a HIP device (no CPU fallback)")
        f = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()
        self.dsm_in, self.dsm_gt, self.orthos = f(dsm_input), f(dsm_target), f(orthos)
        self.h, self.w = self.dsm_in.shape
        self.tile, self.nodata = int(tile_size), float(nodata)
        self.dsm_std, self.ortho_mean, self.ortho_std = float(dsm_std), ortho_mean, float(ortho_std)
        self.device = dev

    def sample(self, positions, pairs=None, aug=None):
        """positions: int [n,2] (y, x); pairs: int [n,V] plane indices into `orthos`; aug: int [n,3] (k, flip_v, flip_h)
        or None.  Returns the DataLoader-shaped batch dict with device tensors."""
        with torch.cuda.device(self.device):       # raw launches go to the current device's stream
            return self._sample(positions, pairs, aug)

    def _sample(self, positions, pairs, aug):
        dev, t = self.device, self.tile
        pos = torch.as_tensor(positions, dtype=torch.int32).reshape(-1, 2)
        if not pos.is_cuda:          # validate on the host (no device->host sync in the training loop)
            if int(pos[:, 0].max()) + t > self.h or int(pos[:, 1].max()) + t > self.w or int(pos.min()) < 0:
                raise ValueError("patch position outside the raster")
        pos = pos.to(dev).contiguous()
        n = pos.shape[0]
        v = 0
        pair_t = omean = None
        if self.orthos is not None and pairs is not None:
            pair_t = torch.as_tensor(pairs, dtype=torch.int32).reshape(n, -1).to(dev).contiguous()
            v = pair_t.shape[1]
        zero = torch.zeros(n, dtype=torch.int32, device=dev)
        sums = torch.empty(n, 2, dtype=torch.float64, device=dev)
        check(load().rd_patch_sums(ptr(self.dsm_in), self.h * self.w, ptr(zero), 1, ptr(pos), n, t, self.w, self.nodata, 1,
                                   ptr(sums), stream_ptr()), "patch_sums")
        dsm_mean = (sums[:, 0] / sums[:, 1]).to(torch.float32)
        if v:
            if self.ortho_mean is None:
                osum = torch.empty(n, 2, dtype=torch.float64, device=dev)
                check(load().rd_patch_sums(ptr(self.orthos), self.h * self.w, ptr(pair_t), v, ptr(pos), n, t, self.w, 0.0, 0,
                                           ptr(osum), stream_ptr()), "patch_sums")
                omean = (osum[:, 0] / osum[:, 1]).to(torch.float32)
            else:
                omean = torch.full((n,), float(self.ortho_mean), dtype=torch.float32, device=dev)
        aug_t = None
        if aug is not None:
            a = torch.as_tensor(aug, dtype=torch.int32).reshape(n, 3)
            aug_t = (a[:, 0] | (a[:, 1] << 2) | (a[:, 2] << 3)).to(torch.int32).to(dev).contiguous()
        inp = torch.empty(n, 1 + v, t, t, dtype=torch.float32, device=dev)
        tgt = msk = None
        if self.dsm_gt is not None:
            tgt = torch.empty(n, 1, t, t, dtype=torch.float32, device=dev)
            msk = torch.empty(n, 1, t, t, dtype=torch.uint8, device=dev)
        check(load().rd_assemble_patches(ptr(self.dsm_in), ptr(self.dsm_gt), ptr(self.orthos) if v else None,
                                         self.h * self.w, ptr(pair_t), v, ptr(pos), ptr(aug_t), ptr(dsm_mean), self.dsm_std,
                                         ptr(omean), self.ortho_std, self.nodata, n, t, self.w, ptr(inp), ptr(tgt), ptr(msk),
                                         stream_ptr()), "assemble_patches")
        batch = {"input": inp, "dsm_mean": dsm_mean, "dsm_std": torch.full((n,), self.dsm_std, device=dev),
                 "patch_offset_x": pos[:, 1], "patch_offset_y": pos[:, 0],
                 "nodata": torch.full((n,), self.nodata, device=dev)}
        if tgt is not None:
            batch["target"], batch["loss_mask"] = tgt, msk.view(torch.bool)
        return batch

    def random_batch(self, n: int, pairs, generator=None, augment: bool = True):
        """n random patch positions (uniform over the raster) + the reference's augmentation draws
        (k in {0..3}, flips with probability 1/2)."""
        g = generator
        ys = torch.randint(0, self.h - self.tile + 1, (n,), generator=g)
        xs = torch.randint(0, self.w - self.tile + 1, (n,), generator=g)
        aug = None
        if augment:
            aug = torch.stack([torch.randint(0, 4, (n,), generator=g), torch.randint(0, 2, (n,), generator=g),
                               torch.randint(0, 2, (n,), generator=g)], 1)
        pair = torch.as_tensor(pairs, dtype=torch.int32)
        if pair.dim() == 1:
            pair = pair.unsqueeze(0).expand(n, -1)
        return self.sample(torch.stack([ys, xs], 1), pair, aug)

    def stream_batches(self, n_batches: int, batch_size: int, pairs, generator=None, augment: bool = True, prefetch: int = 1):
        """Generator of `n_batches` random training batches (the dict of `random_batch`) assembled `prefetch` batches AHEAD on a
        side stream, so that the two assembly kernels of batch k + 1 (0.26 ms for 32 tiles of 256 x 256 x 3) run under batch k's
        forward / backward instead of in front of its own: the consumer's stream waits on one event per batch and the tensors
        are pinned to it with `record_stream`.  The positions / augmentation draws come from `generator` in the same order as
        consecutive `random_batch` calls, so the stream of batches is the same with or without prefetch."""
        side = torch.cuda.Stream(device=self.device)
        queue = []

        def produce():
            with torch.cuda.stream(side):
                b = self.random_batch(batch_size, pairs, generator=generator, augment=augment)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev

        for k in range(n_batches):
            queue.append(produce())
            if len(queue) > max(0, int(prefetch)):
                yield self._hand_over(queue.pop(0))
        while queue:
            yield self._hand_over(queue.pop(0))

    def _hand_over(self, item):
        b, ev = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for v in b.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return b


class SamplerLoader:
    """A `trainloader` for resdepth_amd.Trainer (lib/Trainer.py:61-64,165: anything with `len()` that yields batch dicts) fed by a
    GpuPatchSampler: `n_batches` random augmented batches per epoch, assembled on the GPU one batch ahead (`stream_batches`).
    Takes the place of DataLoader(DsmOrthoDataset(...)) when the rasters fit in HBM (lib/DsmOrthoDataset.py:161-291 does the same
    per-sample work on the CPU).  `generator`: a CPU torch.Generator for the positions / augmentation draws (its state advances
    from epoch to epoch, as a shuffling DataLoader's does)."""

    def __init__(self, sampler: GpuPatchSampler, n_batches: int, batch_size: int, pairs, generator=None, augment: bool = True,
                 prefetch: int = 1):
        self.sampler, self.n_batches, self.batch_size = sampler, int(n_batches), int(batch_size)
        self.pairs, self.generator, self.augment, self.prefetch = pairs, generator, augment, prefetch
        self.dataset = range(self.n_batches * self.batch_size)       # len(loader.dataset), as the Trainer's shard checks read it
        self.drop_last = True

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        return self.sampler.stream_batches(self.n_batches, self.batch_size, self.pairs, generator=self.generator,
                                           augment=self.augment, prefetch=self.prefetch)
