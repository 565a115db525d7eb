"""Tiled full-raster inference with linear blending on the GPU -- the reference's `predict_linear_blend`
(lib/evaluation.py:460-513) with the same signature and return value (np.ndarray [rows, cols] float64).

Differences underneath: tiles go through the HIP engine in batches of any size (the reference uses batch 1 and a
.cpu() round trip per tile, lib/evaluation.py:497), the prediction never leaves the device until its raster rows are
complete, and de-normalisation + blend weights + accumulation are one kernel per batch launched in dataloader order
(fp64 accumulation in the reference's order => run-to-run deterministic, no atomics).

The raster goes back to the host in ROW STRIPES on a copy stream while later tiles are still computing: a stripe leaves
as soon as no remaining tile touches it (the sweep is row-major, lib/rasterutils.py:100-191).  (Measured, N = 1: the copies
run as the runtime's blit kernels at the PCIe rate, about half of their time under the sweep's kernels -- a few tenths of a
per cent of the sweep against r04's single 537 MB copy after the last tile; the design pays off at N > 1.)

Multi-GPU sweep: tiles are sharded by ROW BANDS (`tiling.band_shards`).  A rank's private raster covers its
band only (8192^2 on 8 GPUs: 75 MB instead of 537 MB); the T - stride rows a band shares with the next one go to their
owner point-to-point (8.4 MB per boundary over one xGMI link, added in rank order => deterministic); and every rank
copies the rows it owns straight into ONE host buffer shared by the ranks of the node (POSIX shared memory, page-locked
per rank with rd_host_register) -- eight PCIe links in parallel instead of a full-raster reduce (537 MB ring) + one serial
537 MB device->host copy on rank 0, which capped the r04 design near 6.3x at 8 GPUs.  Datasets that do not expose a band
plan (`shard_plan`), or whose areas go back up the raster, take the r04 route (full-size rasters + reduce to rank 0).
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib, ops
from .tiling import band_shards, regular_grid
from .trainer import DevicePrefetcher

STRIPE_ROWS = 512          # granularity of the streamed device->host copies (8192 columns: 32 MB per stripe)


def _raster_shape(dataset):
    g = getattr(dataset, "dsm_input_gdal", None)
    if g is not None:
        return int(g.RasterYSize), int(g.RasterXSize)
    return tuple(int(v) for v in dataset.raster_shape)


class HostRaster:
    """The host array a sweep delivers.  One process: pinned memory from torch's caching host allocator.  A process group on
    one node: an anonymous shared-memory file (memfd: tmpfs pages, not bound by the size of a /dev/shm mount) created by rank
    0 and mapped by every rank through /proc/<pid>/fd/<n>; each rank page-locks the rows it owns (`register`, kept across
    sweeps that re-use the object) so its device->host copies are asynchronous DMA straight into the shared array.  The
    memory lives as long as a mapping (the returned array) does.  `close()` drops the registration."""

    def __init__(self, rows: int, cols: int, shared: bool):
        self.rows, self.cols, self.shared = int(rows), int(cols), bool(shared)
        self._registered = None
        self.register_error = None
        if not shared:
            self._t = torch.empty((rows, cols), dtype=torch.float64, pin_memory=True)
            self.array = self._t.numpy()
            return
        import torch.distributed as dist
        rank = dist.get_rank()
        name, fd = [None], None
        if rank == 0:
            fd = os.memfd_create("rd_raster")
            os.ftruncate(fd, self.rows * self.cols * 8)
            name[0] = f"/proc/{os.getpid()}/fd/{fd}"
        dist.broadcast_object_list(name, src=0)
        self.array = np.memmap(name[0], dtype=np.float64, mode="r+", shape=(self.rows, self.cols))
        dist.barrier()                      # every rank holds a mapping: the descriptor can go
        if fd is not None:
            os.close(fd)

    def ptr(self, row: int) -> int:
        return self.array.ctypes.data + int(row) * self.cols * 8

    def register(self, row0: int, row1: int) -> None:
        """Page-lock rows [row0, row1) of the shared mapping for this rank's device (whole pages around the range).  A refusal
        (locked-memory limit) is not fatal: the copies then go through the runtime's staging path."""
        if not self.shared or row1 <= row0:
            return
        base = self.array.ctypes.data
        a = max((self.ptr(row0) // 4096) * 4096, base)
        b = min(-(-self.ptr(row1) // 4096) * 4096, base + self.rows * self.cols * 8)
        if self._registered == (a, b):
            return
        self.close()
        try:
            _lib.check(_lib.load().rd_host_register(a, b - a), "host_register")
            self._registered = (a, b)
        except RuntimeError as e:
            self.register_error = str(e)

    def close(self) -> None:
        if self._registered is not None:
            _lib.check(_lib.load().rd_host_unregister(self._registered[0]), "host_unregister")
            self._registered = None


class _StripeCopier:
    """Streams finished rows of a device raster to the host on a copy stream.  `segments`: the row ranges (raster
    coordinates) that may leave before the end of the sweep; `advance(frontier)` copies whatever part of them lies above
    `frontier` (= the first row a remaining tile still touches) once at least STRIPE_ROWS rows are ready."""

    def __init__(self, raster, row_offset, host: HostRaster, segments, device):
        self.raster, self.off, self.host = raster, int(row_offset), host
        self.todo = [[int(a), int(b)] for a, b in segments if b > a]
        self.stream = torch.cuda.Stream(device=device)
        self.device = device

    def _copy(self, a, b):
        cols = self.host.cols
        src = self.raster.data_ptr() + (a - self.off) * cols * 8
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))          # the blends enqueued so far
        self.stream.wait_event(ev)
        _lib.check(_lib.load().rd_copy_to_host_async(self.host.ptr(a), src, (b - a) * cols * 8, self.stream.cuda_stream),
                   "copy_to_host_async")

    def advance(self, frontier, force=False):
        ready = sum(max(0, min(b, frontier) - a) for a, b in self.todo)
        if ready <= 0 or (ready < STRIPE_ROWS and not force):
            return
        rest = []
        for a, b in self.todo:
            e = min(b, frontier)
            if e > a:
                self._copy(a, e)
            if e < b:
                rest.append([max(a, e), b])
        self.todo = rest

    def finish(self, extra=()):
        """Everything that is left (+ `extra` ranges: rows that waited for the exchange), then wait for the copies."""
        for a, b in list(self.todo) + [list(x) for x in extra]:
            if b > a:
                self._copy(a, b)
        self.todo = []
        self.stream.synchronize()


def _frontiers(pos_y, tile_size, rows):
    """frontier[k] = first raster row a tile k, k+1, ... still touches (suffix minimum; `rows` after the last tile)."""
    out = [rows] * (len(pos_y) + 1)
    for k in range(len(pos_y) - 1, -1, -1):
        out[k] = min(out[k + 1], pos_y[k])
    return out


def _local_tile_rows(dataloader):
    """Row (y) of every tile this loader will yield, IN THE ORDER it yields them, from host-side metadata (no device read-back)
    -- or None, which turns the streamed read-back off (everything is copied at the end).  The order is only known when the
    loader walks `dataset.pos` front to back: a SequentialSampler (or no sampler machinery at all) over a dataset with one
    `pos` entry per item.  A shuffled loader, a custom (batch) sampler or a `pos` that means something else would let rows
    leave for the host before every tile that overlaps them has been blended."""
    ds = dataloader.dataset
    pos = getattr(ds, "pos", None)
    if pos is None:
        return None
    sampler = getattr(dataloader, "sampler", None)
    if sampler is not None and not isinstance(sampler, torch.utils.data.SequentialSampler):
        return None
    bs = getattr(dataloader, "batch_sampler", None)
    if bs is not None and not isinstance(bs, torch.utils.data.BatchSampler):
        return None
    try:
        if len(pos) != len(ds):
            return None
        return [int(p[0]) for p in pos]
    except Exception:       # noqa: BLE001
        return None


def _group_on_one_node() -> bool:
    """Do all ranks of the default process group share rank 0's node AND its PID namespace (what HostRaster(shared=True) needs)?
    Every rank reports (boot id, pid-namespace inode); the answer is the same on every rank (an all-gather)."""
    import torch.distributed as dist

    def ident():
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = os.uname().nodename
        try:
            ns = os.readlink("/proc/self/ns/pid")
        except OSError:
            ns = "?"
        return boot + "|" + ns
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, ident())
    return all(o == out[0] for o in out)


def _exchange_overlaps(raster, me, plan, cols, device):
    """Rows of this rank's extent that another rank owns go to their owner; rows this rank owns that other ranks touched come
    in and are added in ascending sender order (fixed order => the same bits every run).  -> the received row ranges."""
    import torch.distributed as dist
    rank = dist.get_rank()
    on_dev = dist.get_backend() == "nccl"
    sends, recvs = [], []
    for r, o in enumerate(plan):
        if r == rank:
            continue
        if me["y0"] is not None:                 # my extent inside r's owned rows
            a, b = max(me["y0"], o["c0"]), min(me["y1"], o["c1"])
            if b > a:
                sends.append((r, a, b))
        if o["y0"] is not None:                  # r's extent inside my owned rows
            a, b = max(o["y0"], me["c0"]), min(o["y1"], me["c1"])
            if b > a:
                recvs.append((r, a, b))
    if not sends and not recvs:
        return []
    ops_, bufs, keep = [], [], []
    for r, a, b in recvs:
        buf = torch.empty((b - a, cols), dtype=torch.float64, device=device if on_dev else "cpu")
        bufs.append(buf)
        ops_.append(dist.P2POp(dist.irecv, buf, r))
    for r, a, b in sends:
        piece = raster[a - me["lo"]:b - me["lo"]]
        if not on_dev:
            piece = piece.cpu()                  # gloo moves host memory; the copy orders itself after the blends
        keep.append(piece)
        ops_.append(dist.P2POp(dist.isend, piece, r))
    for req in dist.batch_isend_irecv(ops_):
        req.wait()
    for (r, a, b), buf in sorted(zip(recvs, bufs), key=lambda t: t[0][0]):
        raster[a - me["lo"]:b - me["lo"]] += buf.to(device, non_blocking=False) if not on_dev else buf
    return [(a, b) for _, a, b in recvs]


_host_cache = {}


def predict_linear_blend(dataloader, model, reduce_to_rank0: bool = True, host=None):
    """-> np.ndarray [rows, cols] float64 (lib/evaluation.py:460-513).  With a process group: every rank sweeps its shard of
    the tiles and the complete raster is returned on rank 0 (band plan: on every rank -- they share the host array);
    `reduce_to_rank0=False` keeps a rank's private, full-size partial raster (no collective).  `host`: None = a fresh host
    array per call (the reference's behaviour); a HostRaster, or "reuse" (one cached HostRaster per raster shape), delivers
    into the same pinned / shared + page-locked memory every time -- for callers that sweep many rasters of one shape and are
    done with a result before the next sweep overwrites it."""
    if not torch.cuda.is_available():
        raise RuntimeError("resdepth_amd.predict_linear_blend runs on a HIP device only (no CPU fallback)")
    is_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
    first = next(model.parameters(), None)
    # the model's own HIP device if it already lives on one, else the current device (the reference moves it to cuda:0)
    device = first.device if first is not None and first.is_cuda else torch.device("cuda", torch.cuda.current_device())
    model.eval()
    model.to(device)
    ds = dataloader.dataset
    rows, cols = _raster_shape(ds)
    tile_size, stride = int(ds.tile_size), int(ds.stride)
    exchange = is_dist and reduce_to_rank0       # world size 1 included: the same code path, nothing to exchange
    plan = getattr(ds, "shard_plan", None) if exchange else None
    banded = bool(plan) and len(plan) == torch.distributed.get_world_size() and plan[0]["monotonic"] \
        and getattr(ds, "shard", (0, 1))[0] == torch.distributed.get_rank()
    if exchange and banded and torch.distributed.get_world_size() > 1 and not _group_on_one_node():
        # the band plan delivers through ONE host array that every rank maps from rank 0's /proc/<pid>/fd/<n> (HostRaster): that
        # path only exists inside rank 0's node and PID namespace.  A group that spans nodes (or containers) takes the dense route
        # (each rank its own raster, one reduce to rank 0)
        banded = False
    if exchange:
        # every rank must take the same route (the dense one ends in a collective)
        flag = torch.tensor([1 if banded else 0], dtype=torch.int32, device=device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        banded = bool(int(flag))
    me = plan[torch.distributed.get_rank()] if banded else {"lo": 0, "hi": rows, "c0": 0, "c1": rows, "y0": 0, "y1": rows}
    lo, hi = me["lo"], me["hi"]
    raster = torch.zeros(max(hi - lo, 0), cols, dtype=torch.float64, device=device)
    reuse = isinstance(host, str) and host == "reuse"
    if reuse:
        host = _host_cache.get((rows, cols, banded))
    own_host = not isinstance(host, HostRaster) or (host.rows, host.cols, host.shared) != (rows, cols, banded)
    if own_host:
        host = HostRaster(rows, cols, shared=banded)
        if reuse:
            _host_cache[(rows, cols, banded)], own_host = host, False
    dense_reduce = exchange and not banded
    copier = None
    tile_rows = _local_tile_rows(dataloader)
    if not dense_reduce:
        # rows that may leave early: the rows this rank delivers, minus the ones other ranks still add to
        early = [(me["c0"], me["c1"])]
        if banded:
            host.register(me["c0"], me["c1"])
            for r, o in enumerate(plan):
                if r != torch.distributed.get_rank() and o["y0"] is not None:
                    a, b = max(o["y0"], me["c0"]), min(o["y1"], me["c1"])
                    if b > a:
                        early = [seg for (s0, s1) in early for seg in ((s0, min(s1, a)), (max(s0, b), s1)) if seg[1] > seg[0]]
        copier = _StripeCopier(raster, lo, host, early, device)
    front = _frontiers(tile_rows, tile_size, rows) if tile_rows is not None else None
    done = 0
    # host batches (a DataLoader's): input / dsm_mean / dsm_std of batch k + 1 go to the device on a copy stream while batch k
    # computes (trainer.DevicePrefetcher; device-resident batches pass through untouched); the per-tile offsets and valid-pixel
    # boxes travel as ONE pinned int32 block per batch, asynchronously -- a pageable `.to(device)` per field would block the host
    # behind everything already enqueued, once per field and batch
    meta_keys = ("patch_offset_y", "patch_offset_x", "patch_valid_pixels_uly", "patch_valid_pixels_ulx", "patch_valid_pixels_lry",
                 "patch_valid_pixels_lrx")
    lo_t = torch.tensor([lo, 0], dtype=torch.int32, device=device) if lo else None
    with torch.no_grad():
        for batch in DevicePrefetcher(dataloader, device, 1):
            x = batch["input"].to(device, non_blocking=True)
            n = x.shape[0]
            y_pred = model(x)
            mean = torch.as_tensor(batch["dsm_mean"]).flatten().to(torch.float32).to(device)
            std = torch.as_tensor(batch["dsm_std"]).flatten().to(torch.float32).to(device)
            cols6 = [torch.as_tensor(batch[k]).flatten() for k in meta_keys]
            if any(c.is_cuda for c in cols6):
                cols6 = [c.to(device) for c in cols6]
                pos = torch.stack(cols6[0:2], 1).to(torch.int32)
                reg = torch.stack(cols6[2:6], 1).to(torch.int32)
            else:           # [pos (n x 2) | reg (n x 4)] in one pinned block, one asynchronous copy, two contiguous views
                blk = torch.cat([torch.stack(cols6[0:2], 1).flatten(), torch.stack(cols6[2:6], 1).flatten()]).to(torch.int32)
                blk = blk.pin_memory().to(device, non_blocking=True)
                pos, reg = blk[:2 * n].view(n, 2), blk[2 * n:].view(n, 4)
            if lo_t is not None:
                pos = pos - lo_t
            if mean.numel() != n or pos.shape[0] != n:
                raise ValueError("batch dict fields must hold one value per tile")
            with _lib.device_of(raster):
                ops.blend_accumulate(y_pred.contiguous(), mean.contiguous(), std.contiguous(), pos.contiguous(),
                                     reg.contiguous(), tile_size, stride, raster)
                done += n
                if copier is not None and front is not None and done < len(front):
                    copier.advance(front[done])
    with _lib.device_of(raster):
        if dense_reduce:
            torch.distributed.reduce(raster, dst=0, op=torch.distributed.ReduceOp.SUM)
            copier = _StripeCopier(raster, 0, host, [(0, rows)], device)
            copier.finish()
        else:
            late = _exchange_overlaps(raster, me, plan, cols, device) if banded else []
            copier.finish(extra=late)
    if banded:
        if own_host:
            host.close()
        torch.distributed.barrier()          # every rank's rows have landed in the shared array
    return host.array


class SyntheticRasterTiles(Dataset):
    """Stand-in for DsmOrthoDataset(sampling_strategy='test') on a synthetic raster (GDAL is out of scope): regular
    grid with stride T/2, per-patch mean centring and a global std as the reference normalises DSM patches
    (lib/DsmOrthoDataset.py:191-203), same sample dict; `shard=(rank, world)` keeps this rank's tiles: a contiguous band of
    tile rows (`shard_mode="bands"`, the default: `tiling.band_shards`, whose plan for all ranks is kept in `shard_plan`
    for `predict_linear_blend`) or every world-th tile (`"stride"`, the r04 round-robin)."""

    def __init__(self, rows: int, cols: int, n_input_channels: int = 3, tile_size: int = 256, stride=None,
                 seed: int = 0, dsm_std: float = 3.0, shard=(0, 1), areas=None, shard_mode: str = "bands"):
        """areas: optional list of ((x0, x1), (y0, y1)) inclusive pixel extents -- the reference sweeps every area of
        its `allowed` regions with its own regular grid (lib/DsmOrthoDataset.py:96-104, lib/rasterutils.py:100-191);
        None = one area covering the whole raster."""
        self.tile_size = int(tile_size)
        self.stride = int(tile_size // 2 if stride is None else stride)
        self.raster_shape = (int(rows), int(cols))
        g = torch.Generator().manual_seed(seed)
        self.raster = torch.randn(n_input_channels, rows, cols, generator=g)
        self.raster[0] = self.raster[0] * dsm_std + 400.0
        self.dsm_std = float(dsm_std)
        if areas is None:
            areas = [((0, cols - 1), (0, rows - 1))]
        pos, reg = regular_grid([a[0] for a in areas], [a[1] for a in areas], self.tile_size, self.stride)
        self.shard, self.shard_plan = (int(shard[0]), int(shard[1])), None
        if shard_mode == "bands":
            self.shard_plan = band_shards(pos, self.tile_size, int(rows), int(shard[1]))
            idx = list(range(self.shard_plan[shard[0]]["i0"], self.shard_plan[shard[0]]["i1"]))
        elif shard_mode in ("bands", "stride"):
            idx = list(range(shard[0], len(pos), shard[1]))
        else:
            raise ValueError("shard_mode must be 'bands' or 'stride'")
        self.pos = [pos[i] for i in idx]
        self.reg = [reg[i] for i in idx]

    def __len__(self):
        return len(self.pos)

    def __getitem__(self, i):
        y, x = self.pos[i]
        t = self.tile_size
        patch = self.raster[:, y:y + t, x:x + t].clone()
        mean = patch[0].double().mean()
        patch[0] = (patch[0] - mean.float()) / self.dsm_std
        uly, ulx, lry, lrx = self.reg[i]
        return {"input": patch, "dsm_mean": mean, "dsm_std": torch.tensor(self.dsm_std),
                "patch_offset_x": torch.tensor(x), "patch_offset_y": torch.tensor(y), "nodata": torch.tensor(-9999.0),
                "patch_valid_pixels_uly": torch.tensor(uly), "patch_valid_pixels_ulx": torch.tensor(ulx),
                "patch_valid_pixels_lry": torch.tensor(lry), "patch_valid_pixels_lrx": torch.tensor(lrx)}
