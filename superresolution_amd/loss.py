"""Loss group of the default trainer configuration, ``core.models.model.LossGroup``
(core/models/model.py:183-207) with ``SSIMLoss`` (core/modules/loss.py:12-29,
core/modules/functional.py:6-9,169-189): ``lambda_l1 * L1 + lambda_ssim * (1 - SSIM)``.

It is the step right after the network in every training step.  ``FusedLossGroup`` computes the
value and its gradient in one HIP call (two tiled separable-filter passes) and can fold the
trainer's pixel clip (core/trainer.py:154-156) in.  The three ``.item()`` synchronisations of the
reference become one optional transfer.  (The torch formulation it is checked against lives with the
other checkers in ``oracle/torch_checkers.py``.)
"""
from typing import Dict, Tuple

import torch
from torch import Tensor, nn


class _FusedL1SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr: Tensor, target: Tensor, lambda_l1: float, lambda_ssim: float, lo: float, hi: float):
        from . import functional as F_
        out3, dsr = F_.l1_ssim_loss(sr, target, lambda_l1, lambda_ssim, lo, hi)
        ctx.dsr = dsr
        ctx.mark_non_differentiable(out3)
        return out3[2].clone(), out3

    @staticmethod
    def backward(ctx, g_total: Tensor, _g_parts):
        dsr, ctx.dsr = ctx.dsr, None
        return dsr * g_total, None, None, None, None, None


class FusedLossGroup(nn.Module):
    """Same interface and value as the reference's ``LossGroup`` on GPU tensors; ``clip=(lo, hi)`` additionally applies
    ``torch.clip(sr, lo, hi)`` (and its backward) inside the kernel, so the caller passes the raw output."""

    def __init__(self, lambda_l1: float, lambda_ssim: float, clip=None, report: bool = True) -> None:
        super().__init__()
        self.lambda_l1, self.lambda_ssim, self.report = lambda_l1, lambda_ssim, report
        self.clip = (float("-inf"), float("inf")) if clip is None else (float(clip[0]), float(clip[1]))

    def forward(self, batch) -> Tuple[Tensor, Dict[str, float]]:
        loss, parts = _FusedL1SSIMFn.apply(batch["sr_frames"], batch["hr_frames"], self.lambda_l1, self.lambda_ssim,
                                           self.clip[0], self.clip[1])
        loss_dict = {}
        if self.report:
            v = parts.tolist()
            loss_dict = {"l1": v[0], "ssim": v[1], "total": v[2]}
        return loss, loss_dict


class _WeightedL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr: Tensor, target: Tensor, weight: Tensor):
        from . import functional as F_
        out, dsr = F_.weighted_l1_loss(sr, target, weight)
        ctx.dsr = dsr
        return out[0]

    @staticmethod
    def backward(ctx, g: Tensor):
        dsr, ctx.dsr = ctx.dsr, None
        return dsr.mul_(g), None, None


def enhance_weighted_l1(sr: Tensor, target: Tensor, enhance_weight: Tensor) -> Tensor:
    """``((1 + nearest_upsample(enhance_weight)) * (sr - target).abs()).mean()`` in one HIP call, value and gradient: the
    pixel-wise loss weight is the temporal network's second output (core/models/model_final.py:499-505), (N,1,h,w) at the
    input resolution; it carries no gradient (it is computed from the inputs alone)."""
    return _WeightedL1Fn.apply(sr, target, enhance_weight.detach())
