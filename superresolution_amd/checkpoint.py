"""Checkpoint interoperability with the reference trainer (core/trainer.py:222-283).

The reference writes ``{"state_dict": {...}, "epoch", "global_step", "model", "optimizer", "scheduler"}`` with
``torch.save`` under the name ``<Model>_<scale>x_last.ckpt`` / ``..._epoch-E_steps-S.ckpt`` and, when loading,
strips the prefix ``module._orig_mod.`` (a DataParallel-wrapped, torch.compile'd model) from every key.  The
modules of this package keep the reference's state-dict keys and OIHW fp32 shapes, so these two functions are
all a user needs to move checkpoints in either direction.  Files are read with ``weights_only=True``: nothing
from the file is executed.
"""
import os
from collections import OrderedDict
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn

_PREFIXES = ("module._orig_mod.", "_orig_mod.", "module.")


def clean_state_dict(ckpt: Dict[str, Any], ignore_prefix: Optional[str] = "module._orig_mod.") -> "OrderedDict[str, torch.Tensor]":
    """core/trainer.py:245-253: unwrap ``{"state_dict": ...}`` and strip the wrapper prefix.  With
    ``ignore_prefix=None`` every known wrapper prefix (DataParallel / DistributedDataParallel / torch.compile, in
    any combination) is stripped."""
    if "state_dict" in ckpt:
        ckpt = ckpt["state_dict"]
    elif "model" in ckpt and isinstance(ckpt["model"], dict):
        ckpt = ckpt["model"]
    out = OrderedDict()
    for key, param in ckpt.items():
        if ignore_prefix is None:
            for p in _PREFIXES:
                if key.startswith(p):
                    key = key[len(p):]
                    break
        elif key.startswith(ignore_prefix):
            key = key[len(ignore_prefix):]
        out[key] = param
    return out


def load_reference_checkpoint(model: nn.Module, ckpt: Union[str, os.PathLike, Dict[str, Any]], strict: bool = True,
                              ignore_prefix: Optional[str] = "module._orig_mod.") -> Tuple[list, list]:
    """Load a checkpoint written by the reference trainer (path or already-loaded dict) into ``model``.
    Returns (missing_keys, unexpected_keys) like ``nn.Module.load_state_dict``."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(os.fspath(ckpt), map_location="cpu", weights_only=True)
    target = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
    result = target.load_state_dict(clean_state_dict(ckpt, ignore_prefix), strict=strict)
    return list(result.missing_keys), list(result.unexpected_keys)


def checkpoint_name(model: nn.Module, sr_scale: int, epoch: Optional[int] = None, global_step: Optional[int] = None) -> str:
    """core/trainer.py:226-229."""
    name = model._get_name()
    if epoch is None:
        return f"{name}_{sr_scale}x_last.ckpt"
    return f"{name}_{sr_scale}x_epoch-{epoch}_steps-{global_step}.ckpt"


def save_reference_checkpoint(model: nn.Module, directory: Union[str, os.PathLike], sr_scale: int, epoch: Optional[int] = None,
                              global_step: Optional[int] = None, optimizer=None, scheduler=None) -> str:
    """Write a checkpoint the reference trainer can resume from (core/trainer.py:222-240, 258-266): CPU tensors under
    ``state_dict``; with an optimizer also ``epoch``, ``global_step``, ``model``, ``optimizer``, ``scheduler`` and the
    ``_last`` file name.  Returns the path.  The reference's ``load_checkpoint`` (core/trainer.py:271-283) evaluates
    ``ckpt["epoch"] + 1`` and feeds ``ckpt["scheduler"]`` to the scheduler, so a resumable file needs all three of
    ``epoch``, ``global_step`` and ``scheduler``: a call with an optimizer but without them raises ``ValueError``."""
    if optimizer is not None and (epoch is None or global_step is None or scheduler is None):
        raise ValueError("save_reference_checkpoint: a resumable checkpoint (optimizer given) needs epoch, global_step and "
                         "scheduler as well; omit the optimizer to write a plain state-dict file")
    target = model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model
    state = OrderedDict((k, v.detach().cpu()) for k, v in target.state_dict().items())
    ckpt: Dict[str, Any] = {"state_dict": state}
    last = optimizer is not None
    if last:
        ckpt.update({"epoch": epoch, "global_step": global_step, "model": state, "optimizer": optimizer.state_dict(),
                     "scheduler": scheduler.state_dict()})
    path = os.path.join(os.fspath(directory), checkpoint_name(target, sr_scale, None if last else epoch, global_step))
    torch.save(ckpt, path)
    return path
