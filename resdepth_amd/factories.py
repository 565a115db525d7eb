"""The factory functions a `train.py` / `test.py`-style driver calls between reading its configuration and
constructing the Trainer (lib/utils.py:275-440), restated over this package's classes so that such a driver can swap
`from lib import utils` for `from resdepth_amd import factories as utils` on the hot path:

    get_loss(cfg)                       lib/utils.py:275-292
    get_model(cfg)                      lib/utils.py:295-316 (+ _collect_model_args :743-753, _count_number_of_input_channels :766-785)
    get_optimizer(cfg, model)           lib/utils.py:318-340   (resdepth_amd.optim; re-exported here)
    get_scheduler(cfg, optimizer)       lib/utils.py:343-377
    get_trainer(cfg, ...)               lib/utils.py:380-440
    valid_tile_size(value, arg_name, min_power)   lib/validate_arguments.py:143-171, called with min_power = depth + 2
                                        (lib/validate_cfg_training.py:640-646, lib/validate_cfg_inference.py:399-406)

`cfg` is whatever the driver holds: the reference uses an EasyDict (not installed here); attribute access and item
access are both accepted (`_field`).  Nothing here touches the GPU.  Deviations, all forced by the toolchain (SURVEY
Appendix B): the `verbose=` keyword the reference passes to the LR schedulers no longer exists in torch >= 2.7 and is
dropped (ReduceLROnPlateau's `verbose=True` only printed a line when the rate changed; the Trainer logs the learning rate
every validated epoch anyway); an unknown name logs the reference's message and then raises the UnboundLocalError the
reference runs into one line later, instead of returning an unbound local.
"""
from __future__ import annotations

import glob
import os
import re
import shutil
import types
from pathlib import Path

import torch

from .optim import get_optimizer  # noqa: F401  (same module-level name as lib/utils.py)

OPTIMIZERS = ["Adam", "SGD"]                                       # lib/arguments.py:53-56
SCHEDULERS = ["ReduceLROnPlateau", "StepLR", "ExponentialLR"]
LOSSES = ["L1"]
ARCHITECTURES = ["UNet"]


def _field(node, name, *default):
    """cfg.a / cfg['a'] -- EasyDict, dict, SimpleNamespace and argparse namespaces all work."""
    if isinstance(node, dict):
        if name in node:
            return node[name]
    elif hasattr(node, name):
        return getattr(node, name)
    if default:
        return default[0]
    raise AttributeError(f"configuration has no field '{name}'")


def _has(node, name):
    return (name in node) if isinstance(node, dict) else hasattr(node, name)


def _as_kwargs(node):
    if node is None:
        return {}
    return dict(node) if isinstance(node, dict) else dict(vars(node))


def _complain(logger, msg, unbound):
    if logger:
        logger.error(msg)
    else:
        print(f"ERROR: {msg}")
    raise UnboundLocalError(f"local variable '{unbound}' referenced before assignment")


def get_loss(cfg, logger=None):
    """cfg.training_settings.loss == 'L1' -> nn.L1Loss(reduction='mean'): the object the Trainer takes as `criterion`
    (resdepth_amd.Trainer evaluates it, together with the de-normalisation of lib/Trainer.py:87-100, as one fused kernel)."""
    name = _field(_field(cfg, "training_settings"), "loss")
    if name == "L1":
        return torch.nn.L1Loss(reduction="mean")
    _complain(logger, f"{name} loss is not implemented. Choose among {LOSSES}.\n", "criterion")


def count_number_of_input_channels(cfg) -> int:
    """'geom' 1 | 'stereo', 'geom-mono' 2 | 'geom-stereo' 3 | 'geom-multiview' views + 1 (lib/utils.py:766-785)."""
    kind = _field(_field(cfg, "model"), "input_channels")
    if kind == "geom":
        return 1
    if kind in ("stereo", "geom-mono"):
        return 2
    if kind == "geom-stereo":
        return 3
    if kind == "geom-multiview":
        return int(re.findall(r"\d+", _field(_field(cfg, "multiview"), "config"))[0]) + 1
    return None         # the reference falls off the end of its if-chain the same way


def collect_model_args(cfg):
    """-> namespace(name, input_channels, settings): `settings` are exactly the UNet constructor's keyword arguments
    (lib/utils.py:743-753)."""
    m = _field(cfg, "model")
    out = types.SimpleNamespace(name=_field(m, "name"), input_channels=_field(m, "input_channels"), settings={})
    if out.name == "UNet":
        out.settings["n_input_channels"] = count_number_of_input_channels(cfg)
        for key in ("start_kernel", "depth", "act_fn_encoder", "act_fn_decoder", "act_fn_bottleneck", "up_mode", "do_BN",
                    "outer_skip", "outer_skip_BN", "bias_conv_layer"):
            out.settings[key] = _field(m, key)
    return out


def get_model(cfg, logger=None):
    """-> (resdepth_amd.UNet, args_model)."""
    from .unet import UNet
    args_model = collect_model_args(cfg)
    if args_model.name == "UNet":
        return UNet(**args_model.settings), args_model
    _complain(logger, f"{args_model.name} model is not implemented. Choose among {ARCHITECTURES}.\n", "model")


def get_scheduler(cfg, optimizer, logger=None):
    """cfg.scheduler.{enabled, name, settings}: ReduceLROnPlateau(mode='min', **settings) | StepLR(**settings) |
    ExponentialLR(**settings); None when disabled."""
    sch = _field(cfg, "scheduler")
    if not _field(sch, "enabled"):
        return None
    name = _field(sch, "name")
    settings = _as_kwargs(_field(sch, "settings", None))
    settings.pop("verbose", None)
    if name == "ReduceLROnPlateau":
        return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", **settings)
    if name == "StepLR":
        return torch.optim.lr_scheduler.StepLR(optimizer, **settings)
    if name == "ExponentialLR":
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, **settings)
    _complain(logger, f"{name} learning rate scheduler is not implemented. Choose among {SCHEDULERS}.\n", "scheduler")


def get_trainer(cfg, trainloader, valloader, model, optimizer, scheduler, criterion):
    """The Trainer's argument object from the configuration (lib/utils.py:380-440), including what a resumed experiment
    inherits from the run it continues: previous tensorboard event files, the training log and Model_best.pth."""
    from .trainer import Trainer
    out = _field(cfg, "output")
    general = _field(cfg, "general")
    config = types.SimpleNamespace(
        trainloader=trainloader, valloader=valloader, model=model, optimizer=optimizer, scheduler=scheduler, criterion=criterion,
        n_epochs=_field(_field(cfg, "training_settings"), "n_epochs"), evaluate_rate=_field(general, "evaluate_rate"),
        save_model_rate=_field(general, "save_model_rate"), freq_average_train_loss=20,
        save_dir=_field(out, "output_directory"), checkpoint_dir=_field(out, "checkpoint_dir"),
        tboard_log_dir=_field(out, "tboard_log_dir"), pretrained_path=None)
    config.log_file = os.path.join(config.save_dir, "training.log")
    os.makedirs(config.tboard_log_dir, exist_ok=True)
    if _has(_field(cfg, "model"), "pretrained_path"):
        config.pretrained_path = _field(_field(cfg, "model"), "pretrained_path")
        experiment = Path(config.pretrained_path).parent.parent
        previous_logs = experiment.parent / "logs" / experiment.name
        if previous_logs.is_dir():
            for event_file in glob.glob(os.path.join(previous_logs, "events.*")):
                shutil.copy(event_file, Path(config.tboard_log_dir) / Path(event_file).name)
        if (experiment / "training.log").is_file():
            os.makedirs(config.save_dir, exist_ok=True)
            shutil.copy(experiment / "training.log", config.log_file)
        best = Path(config.pretrained_path).parent / "Model_best.pth"
        if best.is_file():
            os.makedirs(config.checkpoint_dir, exist_ok=True)
            shutil.copy(best, Path(config.checkpoint_dir) / "Model_best.pth")
    return Trainer(config)


def valid_tile_size(value, arg_name="tile_size", min_power=4, logger=None) -> bool:
    """The reference's tile-size rule: an int among 2^min_power .. 2^11, where its callers pass min_power = depth + 2 ("consistency
    with the number of downsampling layers").  The engine itself accepts more (any height x width that is a multiple of
    2^depth); this is the check a drop-in configuration validator applies."""
    allowed = [2 ** i for i in range(min_power, 12)]
    ok = True
    for bad, msg in ((not isinstance(value, int), "Enter an integer."),
                     (value not in allowed, f"Choose among {allowed}.")):
        if bad:
            text = f"Invalid value for the argument {arg_name}: {value}. {msg}\n"
            if logger:
                logger.error(text)
            else:
                print(f"ERROR: {text}")
            ok = False
    return ok


def validate_tile_size(tile_size, depth, logger=None) -> int:
    """valid_tile_size with the callers' min_power = depth + 2; returns the tile size or raises ValueError."""
    if not valid_tile_size(tile_size, "tile_size", depth + 2, logger):
        raise ValueError(f"tile_size {tile_size!r} is not a power of two in [2^{depth + 2}, 2048] (depth {depth})")
    return tile_size
