"""Does a map that the previous launch wrote (or read) come back from the memory-side cache?  Times the weight-gradient
launch on one image (or a part of one) cold (after a 1 GiB sweep), warm (same operands as the launch before it) and
right after a launch that PRODUCED its dz operand."""
import sys
import torch
from superresolution_amd import functional as F_


def timed(fn, before, reps=6):
    ts = []
    for _ in range(reps):
        before()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    torch.manual_seed(0)
    flush_buf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dw = torch.empty(64, 64, 3, 3, device="cuda")
    db = torch.empty(64, device="cuda")
    wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wp = F_.pack_weights(wt, 64) if hasattr(F_, "pack_weights") else None
    for n, h in ((1, 1024), (1, 512), (1, 256), (2, 1024)):
        x = torch.randn(n, h, 1024, 64, device="cuda").to(torch.bfloat16)
        dz = torch.randn(n, h, 1024, 64, device="cuda").to(torch.bfloat16)
        src = torch.randn(n, h, 1024, 64, device="cuda").to(torch.bfloat16)
        mb = x.numel() * 2 / 2 ** 20

        def flush():
            flush_buf.add_(1)

        def wg():
            F_.conv3x3_wgrad(x, dz, dw, db, 64, 64)

        def warm():
            wg()

        def produce():
            flush()
            torch.add(src, 1.0, out=dz)           # writes dz (reads src)

        def produce_both():
            flush()
            torch.add(src, 1.0, out=dz)
            torch.add(src, 2.0, out=x)

        print(f"n={n} h={h} ({mb:.0f} MiB per map): cold {timed(wg, flush):7.1f} us   same operands again {timed(wg, warm):7.1f} us   "
              f"dz just written {timed(wg, produce):7.1f} us   both just written {timed(wg, produce_both):7.1f} us", flush=True)


if __name__ == "__main__":
    sys.exit(main())
