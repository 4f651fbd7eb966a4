"""Mirrors of the reference's warp / coordinate helpers on libsrhip.so, with the reference's
names, argument order, defaults and error behaviour:

  * ``backward_warping(frame, motion, mode="nearest", align_corners=True)`` -- core/tools/math.py:10-17.
    The committed reference builds a (B,H,W,2) grid and then permutes it as if it were (B,2,H,W),
    which makes ``F.grid_sample`` raise for every real input; this mirror implements the evident
    intent (grid (x, y)-last, no permute), the form all of the function's callers assume.
  * ``backward_warping_history(...)`` -- core/models/model_final.py:17-37 (motion accumulation over
    the history, then warping of frames and buffers).
  * ``make_coord(shape, ranges=None, acquire_batch_dim=True, dtype, device)`` -- core/tools/math.py:107-130.
  * ``dilate_velocity(motion, depth, ksize=3)`` -- core/tools/math.py:80-100 (nothing in the reference calls it; its
    final ``motion[:, y_idx, x_idx, :]`` indexes every sample with every sample's positions and returns
    (B,B,1,H,W,2) instead of the documented shape).  The mirror gathers per sample and returns (B,H,W,2), which for
    B = 1 holds exactly the reference's values.

Tensors are fp32 NCHW on the GPU, as the reference passes them.  The bilinear warp is
differentiable (frame and motion gradients through the scatter-add backward kernel).
"""
from typing import Any, Optional, Tuple

import torch
from torch import Tensor

from . import functional as F_


class _BilinearWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frame: Tensor, motion: Tensor, align_corners: bool) -> Tensor:
        ctx.save_for_backward(frame, motion)
        ctx.align_corners = align_corners
        return F_.backward_warp(frame, motion, "bilinear", align_corners)

    @staticmethod
    def backward(ctx, dy: Tensor):
        frame, motion = ctx.saved_tensors
        dframe, dmotion = F_.backward_warp_bwd(frame, motion, dy.contiguous(), ctx.align_corners)
        return dframe, dmotion, None


def backward_warping(frame: Tensor, motion: Tensor, mode: str = "nearest", align_corners: bool = True) -> Tensor:
    if mode not in ("nearest", "bilinear"):
        raise ValueError(f"backward_warping: unsupported mode {mode!r} (nearest | bilinear)")
    frame, motion = frame.float(), motion.float()
    if mode == "bilinear" and torch.is_grad_enabled() and (frame.requires_grad or motion.requires_grad):
        return _BilinearWarpFn.apply(frame, motion, align_corners)
    return F_.backward_warp(frame.detach(), motion.detach(), mode, align_corners)


def backward_warping_history(motion_vectors: Tensor, history_frames: Tensor, history_buffer: Tensor,
                             mode: str = "bilinear", align_corners: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """motion (B,F,2,h,w), frames (B,F,C,h,w), buffers (B,F,C',h,w) -> warped frames, warped buffers and the
    accumulated motion.  As in the reference, frame f's motion is first warped by, then added to, frame
    f-1's accumulated motion (nearest, align_corners=True), in place."""
    b, f, _, h, w = history_frames.shape
    for idx in range(1, f):
        motion_vectors[:, idx] = backward_warping(motion_vectors[:, idx].contiguous(), motion_vectors[:, idx - 1].contiguous()) \
            + motion_vectors[:, idx - 1]
    mv = motion_vectors.reshape(b * f, -1, h, w)
    wf = backward_warping(history_frames.reshape(b * f, -1, h, w), mv, mode, align_corners)
    wb = backward_warping(history_buffer.reshape(b * f, -1, h, w), mv, mode, align_corners)
    return wf.view(b, f, -1, h, w), wb.view(b, f, -1, h, w), mv.view(b, f, -1, h, w)


def make_coord(shape: Tuple[int, int], ranges: Optional[Any] = None, acquire_batch_dim: bool = True,
               dtype: torch.dtype = torch.float32, device: torch.device = torch.device("cuda")) -> Tensor:
    """(1,2,H,W) (or (2,H,W)) pixel-centre coordinates in [-1,1], channel 0 = x, channel 1 = y."""
    if ranges is not None and tuple(map(tuple, ranges)) != ((-1, 1), (-1, 1)):
        raise NotImplementedError("make_coord on libsrhip covers the default ranges ((-1, 1), (-1, 1)) only")
    if dtype != torch.float32:
        raise TypeError("make_coord: the reference always returns fp32 (it calls .float()); pass dtype=torch.float32")
    h, w = shape
    coord = F_.make_coord(h, w, device)
    return coord if acquire_batch_dim else coord[0]


def dilate_velocity(motion: Tensor, depth: Tensor, ksize: int = 3) -> Tensor:
    """motion (B,H,W,2), depth (B,1,H,W) -> (B,H,W,2): each pixel's motion vector replaced by that of the nearest-depth
    pixel in its ksize x ksize neighbourhood (zero-padded depth, first minimum wins, position clamped to the frame)."""
    return F_.dilate_velocity(motion.float(), depth.float(), ksize)
