#!/usr/bin/env python3
"""Does one operator repeat bit for bit while another process uses the same GPU?  `concurrency_op_check.py [procs] [reps]`
starts `procs` copies that each run the operators below `reps` times on fixed inputs and report where outputs differ."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def noise(seconds):
    """Keeps the card busy with the default network's 4K forward (long MFMA launches, one work-group per CU), so that the
    operators under test share the GPU with a process of another kind."""
    sys.path.insert(0, ROOT)
    import time
    import torch
    from superresolution_amd import SuperSampler
    dev = torch.device("cuda")
    model = SuperSampler().to(dev).eval()
    lr = torch.rand(1, 3, 540, 960, device=dev)
    hb = torch.rand(1, 10, 2160, 3840, device=dev)
    t0 = time.time()
    with torch.no_grad():
        while time.time() - t0 < seconds:
            for _ in range(10):
                model.infer(lr, hb)
            torch.cuda.synchronize()
    return 0


def child(reps):
    sys.path.insert(0, ROOT)
    import torch
    from superresolution_amd import functional as F_
    from superresolution_amd.loss import FusedLossGroup
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    hb = torch.rand(1, 10, 2176, 3840, generator=g).to(dev)
    t48 = torch.randn(1, 544, 960, 48, generator=g).to(torch.bfloat16).to(dev)
    wgt = (torch.randn(3, 3, 3, 3, generator=g) * 0.3).to(dev)
    bias = (torch.randn(3, generator=g) * 0.1).to(dev)
    f32 = torch.relu(torch.randn(1, 540, 960, 32, generator=g)).to(torch.bfloat16).to(dev)
    sr = (torch.rand(4, 3, 1024, 1024, generator=g) * 1.2).to(dev)
    tgt = torch.rand(4, 3, 1024, 1024, generator=g).to(dev)
    loss_fn = FusedLossGroup(1.0, 0.5)

    def loss_grad():
        x = sr.clone().requires_grad_(True)
        total, _ = loss_fn({"sr_frames": x, "hr_frames": tgt})
        total.backward()
        return torch.cat((x.grad.reshape(-1), total.detach().reshape(1)))

    ops = {
        "upsample_int4 (packed fp32 by hand)": lambda: F_.upsample_grid_bilinear_nhwc(f32, 2160, 3840),
        "fused loss fwd+bwd (packed fp32 by hand)": loss_grad,
        "gradient_map+unshuffle": lambda: F_.pixel_unshuffle_to_nhwc_bf16(F_.gradient_map(hb[:, [0, 1, 2, 6, 7, 8]].contiguous(), 1e-6), 4),
        "unshuffle_max": lambda: F_.pixel_unshuffle_to_nhwc_bf16(hb, 4, want_max=True)[0],
        "tail_c3": lambda: F_.pixel_shuffle4_conv3x3_c3(t48, wgt, bias, act=F_.ACT_RELU),
        "shuffle_nhwc16": lambda: F_.pixel_shuffle4_nhwc_to_nhwc16(t48, 3),
    }
    rc = 0
    for name, fn in ops.items():
        first = fn().clone()
        outs = [fn().clone() for _ in range(reps)]
        torch.cuda.synchronize()
        nbad = 0
        for k, o in enumerate(outs):
            if not torch.equal(o.view(torch.int16) if o.dtype == torch.bfloat16 else o, first.view(torch.int16) if first.dtype == torch.bfloat16 else first):
                nbad += 1
                if nbad <= 2:
                    d = (o.float() != first.float()).nonzero()
                    print(f"pid {os.getpid()} {name} rep {k}: {d.shape[0]} elements differ; first {d[0].tolist()} last {d[-1].tolist()}; "
                          f"values {float(o.float()[tuple(d[0])])} vs {float(first.float()[tuple(d[0])])}", flush=True)
        print(f"pid {os.getpid()} {name}: {nbad} of {reps} repetitions differ", flush=True)
        rc |= 1 if nbad else 0
    return rc


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        sys.exit(child(int(sys.argv[2])))
    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        sys.exit(noise(float(sys.argv[2])))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    # the last process is the noise maker when NOISE=<seconds> is set
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(reps)]) for _ in range(procs)]
    nz = None
    if os.environ.get("NOISE"):
        nz = subprocess.Popen([sys.executable, os.path.abspath(__file__), "noise", os.environ["NOISE"]])
    rc = 0
    for p in ps:
        rc |= p.wait()
    if nz is not None:
        nz.kill()
        nz.wait()
    sys.exit(rc)
