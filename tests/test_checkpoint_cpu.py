"""Checkpoint interoperability with the reference trainer's file format (core/trainer.py:222-283)."""
import os

import torch

from superresolution_amd import RepRFN, SuperSampler
from superresolution_amd import checkpoint as C


def test_round_trip_in_the_trainer_format(tmp_path):
    torch.manual_seed(1)
    m = SuperSampler(3, 10, 1)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    p1 = C.save_reference_checkpoint(m, tmp_path, 4, epoch=3, global_step=1200)
    sched = torch.optim.lr_scheduler.StepLR(opt, 10000, 0.98)      # configs/trainer/4x_H2.yaml:18-23
    p2 = C.save_reference_checkpoint(m, tmp_path, 4, epoch=3, global_step=1200, optimizer=opt, scheduler=sched)
    assert os.path.basename(p1) == "SuperSampler_4x_epoch-3_steps-1200.ckpt"      # trainer.py:229
    assert os.path.basename(p2) == "SuperSampler_4x_last.ckpt"                     # trainer.py:227
    raw = torch.load(p2, weights_only=True)
    assert set(raw) == {"state_dict", "epoch", "global_step", "model", "optimizer", "scheduler"} and raw["epoch"] == 3
    torch.manual_seed(2)
    n = SuperSampler(3, 10, 1)
    assert C.load_reference_checkpoint(n, p1) == ([], [])
    for (ka, a), (kb, b) in zip(m.state_dict().items(), n.state_dict().items()):
        assert ka == kb and torch.equal(a, b)


def test_resumable_checkpoint_needs_epoch_step_scheduler_and_resumes(tmp_path):
    """The reference's load_checkpoint (core/trainer.py:271-283) computes ckpt["epoch"] + 1, reads global_step and
    calls scheduler.load_state_dict(ckpt["scheduler"]): a `_last` file without them cannot be resumed, so writing one
    is refused; a complete one goes through the reference's resume logic (restated below)."""
    import pytest
    torch.manual_seed(1)
    m = SuperSampler(3, 10, 1)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, 10000, 0.98)
    for kw in ({}, {"epoch": 2}, {"epoch": 2, "global_step": 7}, {"epoch": 2, "scheduler": sched}):
        with pytest.raises(ValueError):
            C.save_reference_checkpoint(m, tmp_path, 4, optimizer=opt, **kw)
    path = C.save_reference_checkpoint(m, tmp_path, 4, epoch=2, global_step=7, optimizer=opt, scheduler=sched)
    ckpt = torch.load(path, weights_only=True)
    # core/trainer.py:273-281, step by step
    resume_epoch = 3
    assert int(ckpt["epoch"] + 1) == resume_epoch
    assert int(ckpt["global_step"]) == 7
    n = SuperSampler(3, 10, 1)
    n.load_state_dict(ckpt["model"])
    opt2 = torch.optim.AdamW(n.parameters(), lr=1e-4, weight_decay=1e-4)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, 10000, 0.98)
    opt2.load_state_dict(ckpt["optimizer"])
    sched2.load_state_dict(ckpt["scheduler"])
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), n.state_dict().values()))


def test_wrapper_prefixes_are_stripped(ss_golden):
    """A checkpoint of a DataParallel-wrapped, compiled reference model carries `module._orig_mod.` on every key
    (trainer.py:242-253); the golden's reference state dict stands in for it."""
    sd = {k: torch.from_numpy(v) for k, v in ss_golden[1].items()}   # the reference's own state dict (keys and shapes)
    m = SuperSampler(3, 10, 1)
    assert C.load_reference_checkpoint(m, {"state_dict": {"module._orig_mod." + k: v for k, v in sd.items()}}) == ([], [])
    assert all(torch.equal(m.state_dict()[k], v) for k, v in sd.items())
    for prefix in ("module.", "_orig_mod."):
        m2 = SuperSampler(3, 10, 1)
        assert C.load_reference_checkpoint(m2, {"model": {prefix + k: v for k, v in sd.items()}}, ignore_prefix=None) == ([], [])
        assert all(torch.equal(m2.state_dict()[k], v) for k, v in sd.items())
    # the reference's default prefix does not match a bare "module." file: strict loading reports it
    m3 = SuperSampler(3, 10, 1)
    try:
        C.load_reference_checkpoint(m3, {"state_dict": {"module." + k: v for k, v in sd.items()}})
        raise AssertionError("strict load must fail on unstripped keys")
    except RuntimeError:
        pass
    missing, unexpected = C.load_reference_checkpoint(m3, {"state_dict": {"module." + k: v for k, v in sd.items()}}, strict=False)
    assert len(missing) == len(sd) and len(unexpected) == len(sd)


def test_reprfn_checkpoint_both_forms(tmp_path):
    torch.manual_seed(3)
    m = RepRFN()
    p = C.save_reference_checkpoint(m, tmp_path, 4, epoch=0, global_step=10)
    assert os.path.basename(p) == "RepRFN_4x_epoch-0_steps-10.ckpt"
    n = RepRFN()
    assert C.load_reference_checkpoint(n, p) == ([], [])
    d = RepRFN(deploy=True)
    missing, unexpected = C.load_reference_checkpoint(d, p, strict=False)   # train-form file into the deploy form
    assert missing and unexpected
