"""resdepth_amd -- MI355X-native (gfx950) implementation of the ResDepth U-Net training hot path.

Python host surface mirroring the reference (UNet / Trainer / batch dict), hand-written HIP kernels
underneath (libresdepth_hip.so, C ABI in include/resdepth_hip.h).  No CPU fallback.
"""
from .unet import UNet, SkipConnection  # noqa: F401
from .loss import MaskedL1Loss, masked_l1_loss  # noqa: F401
from .optim import FusedAdam, FusedSGD, get_optimizer  # noqa: F401
from .trainer import Trainer, AverageMeter, DevicePrefetcher  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from .plan import PlannedTrainStep  # noqa: F401
from .data import SyntheticDsmOrthoDataset, synthetic_batch  # noqa: F401
from .inference import predict_linear_blend, SyntheticRasterTiles  # noqa: F401
from .sampler import GpuPatchSampler, SamplerLoader  # noqa: F401
from .factories import (get_loss, get_model, get_scheduler, get_trainer, valid_tile_size,  # noqa: F401
                        validate_tile_size)

__all__ = ["UNet", "SkipConnection", "MaskedL1Loss", "masked_l1_loss", "FusedAdam", "FusedSGD", "get_optimizer", "Trainer", "AverageMeter", "DevicePrefetcher", "GraphedTrainStep", "PlannedTrainStep",
           "SyntheticDsmOrthoDataset", "synthetic_batch", "predict_linear_blend", "SyntheticRasterTiles",
           "GpuPatchSampler", "SamplerLoader", "get_loss", "get_model", "get_scheduler", "get_trainer", "valid_tile_size", "validate_tile_size"]
