import torch, time
a = torch.randn(1 << 20, 160, device="cuda").to(torch.bfloat16)
b = torch.randn(1 << 20, 160, device="cuda").to(torch.bfloat16)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
try:
    ms, r = t(lambda: torch.mm(a.t(), b, out_dtype=torch.float32)); print("out_dtype fp32: %.3f ms" % ms, r.dtype)
except Exception as e:
    print("out_dtype failed:", type(e).__name__, str(e)[:100])
ms, r = t(lambda: torch.mm(a.t(), b)); print("bf16 out: %.3f ms" % ms)
ms, r32 = t(lambda: a.float().t() @ b.float()); print("fp32 with conversions: %.3f ms" % ms)
def chunk(step):
    acc = torch.zeros(160, 160, device="cuda")
    for p0 in range(0, a.shape[0], step): acc += torch.mm(a[p0:p0+step].t(), b[p0:p0+step]).float()
    return acc
for st in (1 << 15, 1 << 17, 1 << 18):
    ms, r = t(lambda: chunk(st)); print("chunk %d: %.3f ms rel err %.2e" % (st, ms, float((r - r32).norm() / r32.norm())))
