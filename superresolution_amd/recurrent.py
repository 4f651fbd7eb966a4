"""Default network with the previous output frame fed back: BASELINE configuration "previous-frame output warped by
motion vectors (bilinear reprojection) and concatenated before the conv trunk".

The reference prepares this and stops short of wiring it: its trainer hands every frame of a clip the motion vectors
``velocity`` (B, n_history, 2, th, tw; HR pixel units) and the previous, detached ``sr_frames`` (B, 3, th, tw)
(core/trainer.py:130-137,163), and its model carries the warp hook (``_backward_warping_sequence``,
core/models/model.py:142-145) and the commented motion accumulation (model.py:177-180), but ``forward`` never reads
them.  This module is that wiring on libsrhip.so, and nothing else is new: the module tree, forward contract and
initialisation rule are ``SuperSampler``'s, with ``buffer_channels + frame_channels`` inputs to the G-buffer extractor
(its first layer is (64, 13, 3, 3) for the default sizes).  One fused pass (``functional.warp_concat_nhwc``) warps the
previous frame (bilinear, align_corners=True, zero padding: the arguments of model_final.py:18,30) and writes the
trunk's 16-channel NHWC bf16 input; no fp32 warp result and no concatenation are materialised.

There is no reference network to compare the whole forward with; tests hold it to the composition of the pinned parts
(the oracle's warp and the oracle's network forward on 13 buffer channels).
"""
from typing import Dict, Union

import torch
from torch import Tensor

from . import functional as F_
from .model import SuperSampler


class RecurrentSuperSampler(SuperSampler):
    def __init__(self, frame_channels: int = 3, buffer_channels: int = 10, n_history: int = 1) -> None:
        if n_history != 1:
            # the reference's trainer only fills the previous-output slot for n_history == 1 (core/trainer.py:136-139)
            raise NotImplementedError("RecurrentSuperSampler: n_history must be 1 (one fed-back frame)")
        super().__init__(frame_channels, buffer_channels + frame_channels, n_history)
        self.gbuffer_channels = buffer_channels

    def trunk_input(self, hr_buffers: Tensor, prev_sr: Tensor = None, velocity: Tensor = None) -> Tensor:
        """(B,10,th,tw) G-buffers, previous output (B,3,th,tw) or None, motion (B,2,th,tw) -> (B,th,tw,16) bf16 NHWC."""
        if hr_buffers.shape[1] != self.gbuffer_channels:
            raise ValueError(f"hr_buffers has {hr_buffers.shape[1]} channels, the model was built for {self.gbuffer_channels}")
        if prev_sr is None or velocity is None:
            return F_.warp_concat_nhwc(hr_buffers.float().contiguous(), None, None)
        return F_.warp_concat_nhwc(hr_buffers.float().contiguous(), prev_sr.detach().float().contiguous(),
                                   velocity.detach().float().contiguous(), "bilinear", True)

    def forward(self, batch: Dict[str, Union[Tensor, int]]):
        lr_frames = batch.pop("lr_frames")
        hr_buffers = batch.pop("hr_buffers")
        prev = batch.get("sr_frames")            # previous frame's output; absent for the first frame of a clip
        velocity = batch.get("velocity")         # (B, n_history, 2, th, tw)
        mv = velocity[:, -1] if velocity is not None else None
        g_in = self.trunk_input(hr_buffers[:, -1], prev if isinstance(prev, Tensor) else None, mv)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .training import supersampler_forward_train
            batch["sr_frames"] = supersampler_forward_train(self, lr_frames[:, -1], g_in)
        else:
            batch["sr_frames"] = self.infer(lr_frames[:, -1], g_in)
        return batch
