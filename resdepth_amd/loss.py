"""Masked, de-normalised L1 loss of the reference trainer as one fused HIP reduction
(Trainer._compute_denormalized_loss, lib/Trainer.py:87-100; denormalize_torch,
lib/data_normalization.py:29-38).

    p_i = y_pred_i * std_i + mean_i ; t_i = y_i * std_i + mean_i      (per sample, two roundings)
    both zeroed where loss_mask == 0 ; L1Loss(mean) over all elements ; * numel / sum(mask)

Data-parallel: the normaliser sum(mask) and numel are GLOBAL (all ranks), so the two partial sums are
all-reduced between the reduction and the finishing kernel.
"""
from __future__ import annotations

import torch

from . import _lib, ops


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y, mask_u8, mean, std, grad_sync):
        with _lib.device_of(y_pred):
            sums = ops.masked_l1_partial(y_pred, y, mask_u8, mean, std)
            numel = y_pred.numel()
            if grad_sync is not None:
                numel = grad_sync.allreduce_loss_sums(sums, numel)
            loss, _ = ops.masked_l1_finish(y_pred, y, mask_u8, mean, std, sums, numel, want_loss=True, want_grad=False)
        ctx.save_for_backward(y_pred, y, mask_u8, mean, std, sums)
        ctx.numel = numel
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        y_pred, y, mask_u8, mean, std, sums = ctx.saved_tensors
        # gout stays on the device: the kernel reads it through a pointer (no host sync)
        g = gout.detach().to(torch.float32).contiguous()
        with _lib.device_of(y_pred):
            _, dyp = ops.masked_l1_finish(y_pred, y, mask_u8, mean, std, sums, ctx.numel, gout=g, want_loss=False,
                                          want_grad=True)
        return dyp, None, None, None, None, None


def _to_device_f32(v, dev):
    """Host values go through a pinned staging buffer and an ASYNCHRONOUS copy: a pageable `.to(device)` would block the
    host until everything already enqueued on the stream (the whole forward) has finished."""
    t = torch.as_tensor(v).flatten().to(torch.float32)
    if not t.is_cuda:
        t = t.pin_memory().to(dev, non_blocking=True)
    return t.contiguous()


def _prep(y_pred, y, loss_mask, mean, std):
    dev = y_pred.device
    if not y_pred.is_cuda:
        raise RuntimeError("resdepth_amd.masked_l1_loss runs on a HIP device only (no CPU fallback)")
    n = y_pred.shape[0]
    y = y.to(dev, torch.float32).contiguous()
    m = loss_mask.to(dev)
    if m.dtype != torch.uint8:
        m = (m != 0).to(torch.uint8) if m.dtype != torch.bool else m.view(torch.uint8)
    # mean/std arrive as CPU tensors in the reference (batch['dsm_mean'|'dsm_std'], lib/Trainer.py:174-175) and
    # are converted with .tolist() -> python floats -> fp32 scalars
    mean32, std32 = _to_device_f32(mean, dev), _to_device_f32(std, dev)
    if mean32.numel() != n or std32.numel() != n:
        raise ValueError("dsm_mean / dsm_std must hold one value per sample")
    return y_pred.contiguous(), y, m.contiguous(), mean32, std32


def masked_l1_loss(y_pred, y, loss_mask, mean, std, grad_sync=None):
    """Functional form; returns a 0-dim device tensor with autograd to y_pred."""
    yp, y, m, mean32, std32 = _prep(y_pred, y, loss_mask, mean, std)
    return _MaskedL1.apply(yp, y, m, mean32, std32, grad_sync)


class MaskedL1Loss(torch.nn.Module):
    """Module form (drop-in for the `criterion` + denormalisation pair of the reference Trainer)."""

    def __init__(self, grad_sync=None):
        super().__init__()
        self.grad_sync = grad_sync

    def forward(self, y_pred, y, loss_mask, mean, std):
        return masked_l1_loss(y_pred, y, loss_mask, mean, std, self.grad_sync)
