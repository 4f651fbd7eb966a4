#!/usr/bin/env python3
"""HIP-graph capture of the launch-bound RepRFN passes: eager vs replay time, and equality of the results."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import RepRFN
from superresolution_amd.dp import FlatGradAllReducer

dev = torch.device("cuda")


def timeit(fn, n):
    gc.collect(); gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def infer():
    torch.manual_seed(42)
    m = RepRFN(deploy=True).to(dev).eval()
    x = torch.rand(1, 3, 540, 960, device=dev)
    with torch.no_grad():
        ref = m(x).clone()
        for _ in range(3):
            m(x)
        eager = timeit(lambda: m(x), 50)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                m(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x)
        x2 = torch.rand(1, 3, 540, 960, device=dev)
        ref2 = None
        g.replay()
        torch.cuda.synchronize()
        print("infer: replay == eager:", torch.equal(out, ref))
        replay = timeit(g.replay, 50)
        x.copy_(x2); g.replay(); torch.cuda.synchronize()
        got2 = out.clone()
        ref2 = m(x2)
        print("infer: replay on new input == eager:", torch.equal(got2, ref2))
    print(f"infer 540x960 -> 4K: eager {eager:.3f} ms, graph replay {replay:.3f} ms")


def train():
    torch.manual_seed(42)
    model = RepRFN().to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    red = FlatGradAllReducer(params)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True, capturable=True)
    x = torch.rand(16, 3, 256, 256, device=dev)
    target = torch.rand(16, 3, 1024, 1024, device=dev)

    def step():
        red.zero_grad()
        loss = (model(x) - target).abs().mean()
        loss.backward()
        red.all_reduce()
        opt.step()
        return loss.detach()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            l0 = step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    eager = timeit(step, 20)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    losses = []
    for _ in range(5):
        g.replay()
        losses.append(float(loss))
    print("train: losses under replay:", ["%.5f" % v for v in losses], "first eager %.5f" % float(l0))
    replay = timeit(g.replay, 20)
    print(f"train step: eager {eager:.3f} ms, graph replay {replay:.3f} ms; finite grads: {bool(torch.isfinite(red.flat).all())}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("infer", "both"):
        infer()
    if which in ("train", "both"):
        train()
