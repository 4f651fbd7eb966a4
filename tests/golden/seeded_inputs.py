"""Test inputs that are functions of seeds alone, shared by the golden generator (tests/golden/make_golden.py, which feeds
them to the reference) and the GPU tests (which feed them to the HIP path): nothing is stored, both sides build the same
tensors from torch's CPU generator."""
import torch


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def rnd(gen, *shape, lo=-1.0, hi=1.0):
    return bf16_round(torch.rand(*shape, generator=gen) * (hi - lo) + lo)


def temporal_train_inputs(h=48, w=64, s=4, seed=91):
    """Inputs of the temporal training golden, from seeds alone (the GPU test calls this very function): one clip with
    one history frame at LR h x w."""
    from torch.nn import functional as F
    g = torch.Generator().manual_seed(seed)
    b, f = 1, 1
    lr_frame = rnd(g, b, 3, h, w, lo=0.0, hi=1.0)
    hr_buffer = rnd(g, b, 10, h * s, w * s, lo=0.0, hi=1.0)
    hr_buffer[:, 9] *= 5.0                       # depth beyond the first clamp threshold
    lr_buffer = F.interpolate(hr_buffer, scale_factor=1.0 / s, mode="nearest")
    motion = rnd(g, b, f, 2, h, w, lo=-2.0, hi=2.0)
    hist_frames = (lr_frame[:, None] + 0.05 * rnd(g, b, f, 3, h, w)).clamp(0, 1)
    hist_buffer = (lr_buffer[:, None] + 0.02 * rnd(g, b, f, 10, h, w)).clamp(0, 6)
    target = rnd(g, b, 3, h * s, w * s, lo=0.0, hi=1.0)
    G = rnd(g, b, 3, h * s, w * s, lo=0.5, hi=1.5)
    return [lr_frame, hr_buffer, lr_buffer, motion, hist_frames, hist_buffer], target, G
