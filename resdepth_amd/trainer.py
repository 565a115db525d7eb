"""Trainer with the reference's surface (lib/Trainer.py): `Trainer(args)` where `args` carries
model, optimizer, scheduler, criterion, trainloader, valloader, n_epochs, evaluate_rate, save_model_rate,
freq_average_train_loss, save_dir, log_file, checkpoint_dir, tboard_log_dir, pretrained_path
(lib/utils.py:395-439); methods `train()`, `inference_one_epoch(epoch, phase)`,
`inference_one_batch(batch, phase) -> {'MAE_metric': float}`; checkpoints
{'epoch','model_state_dict','optimizer_state_dict','loss_train','loss_val'[,'scheduler_state_dict']} written as
Model_best.pth / Model_last.pth / Model_after_{k}_epochs.pth (lib/Trainer.py:145-157,283-317).

What is different underneath:
  * forward/backward run on the HIP engine; the criterion + de-normalisation pair (lib/Trainer.py:87-100) is the
    fused `masked_l1_loss` (the `criterion` argument must be an nn.L1Loss(reduction='mean'), the only loss the
    reference offers, lib/utils.py:285);
  * the epoch loop keeps the per-step loss on the device and reads it back only when a value is logged
    (every `freq_average_train_loss` iterations and at epoch end) instead of one host sync per step
    (lib/Trainer.py:197); `inference_one_batch` still returns a python float as the reference does;
  * data parallel: with a torch.distributed process group initialised every rank trains on its own loader
    shard; gradients, the loss normaliser (and optionally BN statistics) are exchanged by resdepth_amd.dp;
    rank 0 alone logs and writes checkpoints;
  * tensorboard is optional (a no-op writer is used when the package is absent).
"""
from __future__ import annotations

import logging
import math
import os
import time

import torch

from .loss import masked_l1_loss


def _get(args, name, default=None):
    if isinstance(args, dict):
        return args.get(name, default)
    return getattr(args, name, default)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_hparams(self, *a, **k):
        pass

    def close(self):
        pass


def _make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433
        return SummaryWriter(log_dir=log_dir)
    except Exception:  # tensorboard not installed
        return _NullWriter()


class AverageMeter:
    """Running mean of a scalar (same fields as lib/AverageMeter.py: val, avg, sum, count)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class _DeviceMeter:
    """Accumulates device scalars without synchronising; `flush_into(meter)` does one read-back."""

    def __init__(self):
        self.pending = []

    def add(self, t):
        self.pending.append(t.detach().reshape(()))

    def flush_into(self, meter: AverageMeter):
        if self.pending:
            vals = torch.stack(self.pending).double().cpu().tolist()
            self.pending = []
            for v in vals:
                meter.update(float(v))


class DevicePrefetcher:
    """Host -> device staging of batch k + 1 on a COPY stream while batch k computes (the reference moves each batch
    with three blocking `.to(device)` calls in front of the forward, lib/Trainer.py:165-168; its loaders are pinned,
    train.py:146-161).  Per batch: `input` / `target` / `loss_mask` (35.6 MB at cfg-S batch 32, ~0.6-0.7 ms over PCIe
    Gen5) and the per-sample `dsm_mean` / `dsm_std` go over as asynchronous copies from pinned memory; the consumer's
    stream waits on ONE event per batch, and the device tensors are pinned to it with `record_stream` (their blocks go
    back to the caching allocator only after the step that read them).  Host tensors that are not pinned yet are copied
    into pinned staging buffers first (torch's caching host allocator recycles them; a pageable non_blocking copy
    would stall the host until the stream drains).  Tensors already on the device pass through untouched.

    Iterating it yields the loader's batch dicts with those five fields replaced by device tensors; everything else
    (offsets, nodata, valid-pixel boxes) is handed on as the loader produced it."""

    FIELDS = ("input", "target", "loss_mask")
    SCALARS = ("dsm_mean", "dsm_std")

    def __init__(self, loader, device, depth: int = 1):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        out = dict(batch)
        moved = []
        with torch.cuda.stream(self.copy_stream):
            for k in self.FIELDS + self.SCALARS:
                v = batch.get(k) if isinstance(batch, dict) else None
                if v is None:
                    continue
                t = src = torch.as_tensor(v)
                if k in self.SCALARS and not src.is_cuda:
                    # what masked_l1_loss feeds its kernels (lib/Trainer.py:174-175); device-resident scalars are left to the
                    # consumer (a conversion kernel launched HERE would run on the copy stream)
                    t = t.flatten().to(torch.float32)
                if not t.is_cuda:
                    if not t.is_pinned():
                        t = t.pin_memory()
                    t = t.to(self.device, non_blocking=True)
                if t.is_cuda and t is not src:
                    moved.append(t)          # born on the copy stream: pinned to the consumer's stream at hand-over
                out[k] = t
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, moved, ev

    def __iter__(self):
        queue = []
        for batch in self.loader:
            queue.append(self._stage(batch))
            if len(queue) > self.depth:
                yield self._hand_over(queue.pop(0))
        while queue:
            yield self._hand_over(queue.pop(0))

    def _hand_over(self, staged):
        out, moved, ev = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in moved:
            t.record_stream(cur)
        return out


class Trainer:
    def __init__(self, args):
        self.config = args
        self.save_dir = _get(args, "save_dir")
        self.checkpoint_dir = _get(args, "checkpoint_dir")
        self.tboard_log_dir = _get(args, "tboard_log_dir")
        self.pretrained_path = _get(args, "pretrained_path")
        self.log_file = _get(args, "log_file")
        self.is_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.rank = torch.distributed.get_rank() if self.is_dist else 0
        self.is_main = self.rank == 0
        for d in (self.save_dir, self.checkpoint_dir):
            if d and self.is_main:
                os.makedirs(d, exist_ok=True)
        self.path_model_best = os.path.join(self.checkpoint_dir, "Model_best.pth")
        self.path_model_last = os.path.join(self.checkpoint_dir, "Model_last.pth")
        self.writer = _make_writer(self.tboard_log_dir) if self.is_main else _NullWriter()
        self.logger = self._setup_logger()

        self.start_epoch = 0
        self.n_epochs = _get(args, "n_epochs")
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", "0")) if self.is_dist else 0
            self.device = torch.device("cuda", local)
            torch.cuda.set_device(self.device)     # raw kernel launches and RCCL use the current device
        else:
            self.device = torch.device("cpu")
        self.model = _get(args, "model")
        self.optimizer = _get(args, "optimizer")
        self.scheduler = _get(args, "scheduler")
        self.criterion = _get(args, "criterion")
        if self.criterion is not None and not (isinstance(self.criterion, torch.nn.L1Loss)
                                                and self.criterion.reduction == "mean"):
            raise NotImplementedError("resdepth_amd.Trainer implements the reference's L1 (mean) criterion only")
        self.evaluate_rate = _get(args, "evaluate_rate", 1)
        self.save_model_rate = _get(args, "save_model_rate", 10 ** 9)
        self.freq_average_train_loss = _get(args, "freq_average_train_loss", 20)
        self.best_loss = math.inf
        self.index_best_loss = math.inf
        self.grad_sync = getattr(self.model, "grad_sync", None)
        # host -> device staging of the next batch under the current step (DevicePrefetcher); 0 = the copies ride the compute
        # stream in front of the forward, as in r04
        self.prefetch_batches = int(_get(args, "prefetch_batches", 1))
        # hip_graph: replay the training iteration as ONE captured HIP graph (resdepth_amd.graph.GraphedTrainStep) instead of
        # ~110 launches -- for launch-bound iterations (cfg-S: batch <= 6; +39 % at batch 4), bit-identical; off by default
        # (at the benchmark batch the eager two-stream backward is 4-5 % faster).  Ragged batches, validation and
        # multi-rank gradient synchronisation run the eager iteration either way.
        self.hip_graph = bool(_get(args, "hip_graph", False))
        # launch_plan (r06): replay the training iteration from a recorded launch plan (resdepth_amd.plan.PlannedTrainStep) -- the
        # same ~110 kernels enqueued from C on the real two streams: the eager iteration's GPU time (two-stream overlap kept,
        # bit-identical) at a fraction of its host time, and it works under data parallelism (the collectives are issued between
        # the plan's segments).  Off by default in the Trainer shell (a drop-in loop should not change behaviour silently).
        self.launch_plan = bool(_get(args, "launch_plan", False))
        self._graphed = None

        if self.pretrained_path is not None:
            self._load_pretrain(self.pretrained_path)
        else:
            self.logger.info("\nStart training from scratch.\n")
            self.model = self.model.to(self.device)
        self.loader = {"train": _get(args, "trainloader"), "val": _get(args, "valloader")}

        first = next(iter(self.loader["train"]))          # the reference also draws one batch here (lib/Trainer.py:61-64)
        self.batch_size = first["input"].shape[0]
        if self.is_dist and self.grad_sync is not None:
            # every step issues collectives (loss normaliser, gradients, SyncBN): ranks with a different number of
            # batches would deadlock, a different batch size would bias the SyncBN statistics
            for phase in ("train", "val"):
                if self.loader[phase] is not None:
                    self.grad_sync.check_equal_across_ranks(len(self.loader[phase]), f"len({phase} loader)")
            self.grad_sync.check_equal_across_ranks(self.batch_size, "batch size")
            # equal loader lengths + equal first-batch size leave one way for per-rank batch sizes to differ: a ragged LAST
            # batch (drop_last=False).  Its size is what has to agree -- shards of different length that yield the same
            # number of equally sized batches (drop_last=True) are legitimate
            ds = getattr(self.loader["train"], "dataset", None)
            if ds is not None and hasattr(ds, "__len__") and not getattr(self.loader["train"], "drop_last", False):
                self.grad_sync.check_equal_across_ranks(len(ds) % self.batch_size, "size of the last (ragged) train batch")
        self.hparams = {"batch_size": self.batch_size, "lr_initial": self._get_lr(),
                        "optimizer": type(self.optimizer).__name__, "scheduler": "None", "patience": -1, "step_size": -1}
        if self.scheduler is not None:
            self.hparams["scheduler"] = type(self.scheduler).__name__
            if self.hparams["scheduler"] == "ReduceLROnPlateau":
                self.hparams["patience"] = self.scheduler.patience
            elif self.hparams["scheduler"] == "StepLR":
                self.hparams["step_size"] = self.scheduler.step_size

    # ------------------------------------------------------------------------------------------
    def _setup_logger(self):
        logger = logging.getLogger(f"resdepth_amd.train.{id(self)}")
        logger.setLevel(logging.INFO)
        logger.propagate = False
        if self.is_main:
            logger.addHandler(logging.StreamHandler())
            if self.log_file:
                os.makedirs(os.path.dirname(self.log_file) or ".", exist_ok=True)
                logger.addHandler(logging.FileHandler(self.log_file))
        else:
            logger.addHandler(logging.NullHandler())
        return logger

    def _get_lr(self, group=0):
        return self.optimizer.param_groups[group]["lr"]

    @staticmethod
    def _extract_inputs_outputs_loss_masks(batch):
        return batch["input"], batch["target"], batch["loss_mask"]

    def _load_pretrain(self, resume):
        if not os.path.isfile(resume):
            raise ValueError(f"No checkpoint found at '{resume}.\n'")
        ckpt = torch.load(resume, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.model = self.model.to(self.device)          # model first, then the optimizer state (lib/Trainer.py:122-126)
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if "scheduler_state_dict" in ckpt and self.scheduler is not None:
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.start_epoch = ckpt["epoch"] + 1
        self.n_epochs += self.start_epoch
        self.best_loss = ckpt["loss_val"]
        self.index_best_loss = ckpt["epoch"]
        self.logger.info(f"\n\nRestoring the pretrained model from epoch {self.start_epoch}.")
        self.logger.info(f"Successfully load pretrained model from {resume}!\n")
        self.logger.info(f"Current best loss {self.best_loss}\n")

    def _save_checkpoint(self, epoch, loss_train, loss_val, filepath):
        if not self.is_main:
            return
        state = {"epoch": epoch, "model_state_dict": self.model.state_dict(),
                 "optimizer_state_dict": self.optimizer.state_dict(), "loss_train": loss_train, "loss_val": loss_val}
        if self.scheduler is not None:
            state["scheduler_state_dict"] = self.scheduler.state_dict()
        torch.save(state, filepath)

    # ------------------------------------------------------------------------------------------
    def _loss_on_device(self, batch, train: bool):
        """forward (+ backward when training); returns the 0-dim device loss (no host sync)."""
        x, y, loss_mask = self._extract_inputs_outputs_loss_masks(batch)
        x = x.to(self.device, non_blocking=True)
        y = y.to(self.device, non_blocking=True)
        loss_mask = loss_mask.to(self.device, non_blocking=True)
        mean = torch.flatten(batch["dsm_mean"])
        std = torch.flatten(batch["dsm_std"])
        if train:
            self.model.train()
            y_pred = self.model(x)
            loss = masked_l1_loss(y_pred, y, loss_mask, mean, std, grad_sync=self.grad_sync)
            loss.backward()
        else:
            self.model.eval()
            with torch.no_grad():
                y_pred = self.model(x)
                loss = masked_l1_loss(y_pred, y, loss_mask, mean, std, grad_sync=self.grad_sync)
        return loss.detach()

    def inference_one_batch(self, batch, phase):
        assert phase in ["train", "val"]
        loss = self._loss_on_device(batch, phase == "train")
        return {"MAE_metric": float(loss.item())}

    @staticmethod
    def stats_dict():
        return {"MAE_metric": 0.0}

    def stats_meter(self):
        return {k: AverageMeter() for k in self.stats_dict()}

    def inference_one_epoch(self, epoch, phase):
        assert phase in ["train", "val"]
        meters = self.stats_meter()
        dev = _DeviceMeter()
        loader = self.loader[phase]
        num_iter = len(loader)
        if self.prefetch_batches > 0 and self.device.type == "cuda":
            loader = DevicePrefetcher(loader, self.device, self.prefetch_batches)
        params = list(self.model.parameters())
        for p in params:
            p.grad = None
        graphed = None
        if phase == "train" and (self.hip_graph or self.launch_plan) and self.device.type == "cuda":
            if self._graphed is None or self._graphed.model is not self.model or self._graphed.optimizer is not self.optimizer:
                if self.launch_plan:
                    from .plan import PlannedTrainStep
                    self._graphed = PlannedTrainStep(self.model, self.optimizer)
                else:
                    from .graph import GraphedTrainStep
                    self._graphed = GraphedTrainStep(self.model, self.optimizer)
            graphed = self._graphed
            self.model.train()
        for c_iter, batch in enumerate(loader):
            if graphed is not None:
                x, y, loss_mask = self._extract_inputs_outputs_loss_masks(batch)
                to = lambda t, dt=None: t.to(self.device, dtype=dt, non_blocking=True)      # noqa: E731
                # .clone(): after a replay the loss is the graph's own output buffer, rewritten by the next one
                dev.add(graphed(to(x), to(y), to(loss_mask), to(torch.flatten(batch["dsm_mean"]), torch.float32),
                                to(torch.flatten(batch["dsm_std"]), torch.float32)).clone())
            else:
                dev.add(self._loss_on_device(batch, phase == "train"))
            if phase == "train":
                if graphed is None:
                    self.optimizer.step()
                    for p in params:
                        p.grad = None
                if (c_iter + 1) % self.freq_average_train_loss == 0:
                    dev.flush_into(meters["MAE_metric"])
                    curr_iter = num_iter * epoch + (c_iter + 1)
                    message = f"{phase}:\tEpoch: {epoch} [{c_iter + 1}/{num_iter}]\t"
                    for key, value in meters.items():
                        self.writer.add_scalar(f"train/{key}", value.avg, curr_iter)
                        message += f"{key}: {value.avg:.6f}\t"
                        value.reset()
                    self.logger.info(message)
                    self.writer.add_scalar("train/learning_rate", self._get_lr(), curr_iter)
        dev.flush_into(meters["MAE_metric"])
        return meters

    def train(self):
        self.logger.info("Start training...\n")
        t0 = time.time()
        train_meter = val_meter = None
        epoch = self.start_epoch - 1
        for epoch in range(self.start_epoch, self.n_epochs):
            head = f"Epoch {epoch}/{self.n_epochs - 1}"
            self.logger.info("\n{}\n{}\n".format(head, "-" * len(head)))
            train_meter = self.inference_one_epoch(epoch, "train")
            if (epoch + 1) % self.evaluate_rate == 0:
                val_meter = self.inference_one_epoch(epoch, "val")
                message = f"\nval:\tEpoch: {epoch}\t\t"
                for key, value in val_meter.items():
                    self.writer.add_scalar(f"val/{key}", value.avg, epoch)
                    message += f"{key}: {value.avg:.6f}\t"
                self.logger.info(message + "\n")
                self.writer.add_scalar("val/learning_rate", self._get_lr(), epoch)
                v = val_meter["MAE_metric"].avg
                if v < self.best_loss:
                    self.best_loss, self.index_best_loss = v, epoch
                    self._save_checkpoint(epoch, train_meter["MAE_metric"].avg, v, self.path_model_best)
                    self.writer.add_hparams(hparam_dict=self.hparams, metric_dict={"hparam/MAE_metric": v},
                                            run_name=self.tboard_log_dir)
                if self.scheduler is not None:
                    if type(self.scheduler).__name__ == "ReduceLROnPlateau":
                        self.scheduler.step(v)
                    else:
                        self.scheduler.step()
            if (epoch + 1) % self.save_model_rate == 0 and epoch > self.evaluate_rate and val_meter is not None:
                name = "Model_after_" + str(epoch + 1) + "_epochs.pth"
                self._save_checkpoint(epoch, train_meter["MAE_metric"].avg, val_meter["MAE_metric"].avg,
                                      os.path.join(self.checkpoint_dir, name))
        elapsed = time.strftime("%H:%M:%S", time.gmtime(time.time() - t0))
        self.logger.info(f"\n\nTraining finished!\nTraining time: {elapsed}")
        self.logger.info(f"\nBest model at epoch: {self.index_best_loss}")
        self.logger.info("Validation loss of the best model: {:.6f}".format(self.best_loss))
        self.writer.close()
        if train_meter is not None:
            self._save_checkpoint(epoch, train_meter["MAE_metric"].avg,
                                  val_meter["MAE_metric"].avg if val_meter is not None else math.inf,
                                  self.path_model_last)
