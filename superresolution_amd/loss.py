"""Loss group of the default trainer configuration, ``core.models.model.LossGroup``
(core/models/model.py:183-207) with ``SSIMLoss`` (core/modules/loss.py:12-29,
core/modules/functional.py:6-9,169-189): ``lambda_l1 * L1 + lambda_ssim * (1 - SSIM)``.

It is the step right after the network in every training step.  ``LossGroup`` is the reference's
formulation on torch operators (five depthwise 11x11 Gaussian filters + an element-wise map) and
runs anywhere; ``FusedLossGroup`` computes the same value and its gradient in one HIP call
(two tiled separable-filter passes) and can fold the trainer's pixel clip
(core/trainer.py:154-156) in.  The three ``.item()`` synchronisations of the reference become one
optional transfer.
"""
from typing import Dict, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F


def _gaussian(window_size: int, sigma: float) -> Tensor:
    coord = torch.arange(window_size, dtype=torch.float32)
    gauss = torch.exp(-(coord - window_size // 2) ** 2 / (2 * sigma ** 2))
    return gauss / gauss.sum()


class SSIMLoss(nn.Module):
    def __init__(self, window_size: int = 11, channel: int = 3, size_average: bool = True) -> None:
        super().__init__()
        self.window_size, self.channel, self.size_average = window_size, channel, size_average
        w1 = _gaussian(window_size, 1.5).view(window_size, 1)
        w2 = torch.mm(w1, w1.t()).view(1, 1, window_size, window_size)
        self.register_buffer("window", w2.expand(channel, 1, window_size, window_size).contiguous())

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        assert x.shape == y.shape, "Two tensor must have the same shape."
        assert x.shape[1] == self.channel, "Input and gaussian window must have the same channels."
        pad, ch, win = self.window_size // 2, self.channel, self.window
        mu_x = F.conv2d(x, win, padding=pad, groups=ch)
        mu_y = F.conv2d(y, win, padding=pad, groups=ch)
        mu_x2, mu_y2, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
        sigma_x2 = F.conv2d(x * x, win, padding=pad, groups=ch) - mu_x2
        sigma_y2 = F.conv2d(y * y, win, padding=pad, groups=ch) - mu_y2
        sigma_xy = F.conv2d(x * y, win, padding=pad, groups=ch) - mu_xy
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        ssim_map = ((2 * mu_xy + c1) * (2 * sigma_xy + c2)) / ((mu_x2 + mu_y2 + c1) * (sigma_x2 + sigma_y2 + c2))
        return 1 - (ssim_map.mean() if self.size_average else ssim_map)


class LossGroup(nn.Module):
    def __init__(self, lambda_l1: float, lambda_ssim: float, report: bool = True) -> None:
        super().__init__()
        self.l1 = nn.L1Loss()
        self.ssim = SSIMLoss()
        self.lambda_l1, self.lambda_ssim, self.report = lambda_l1, lambda_ssim, report

    def forward(self, batch) -> Tuple[Tensor, Dict[str, float]]:
        target, output = batch["hr_frames"], batch["sr_frames"]
        l1 = self.l1(output, target)
        ssim = self.ssim(output, target)
        loss = self.lambda_l1 * l1 + self.lambda_ssim * ssim
        loss_dict = {}
        if self.report:  # one device->host transfer instead of three
            v = torch.stack((l1.detach(), ssim.detach(), loss.detach())).tolist()
            loss_dict = {"l1": v[0], "ssim": v[1], "total": v[2]}
        return loss, loss_dict


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
    """Same interface and value as ``LossGroup`` on GPU tensors; ``clip=(lo, hi)`` additionally applies
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
