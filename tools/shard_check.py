import sys, torch
sys.path.insert(0, ".")
from superresolution_amd import SuperSampler, functional as F_
torch.manual_seed(42)
m = SuperSampler(3, 10, 1).cuda().eval()
g = torch.Generator().manual_seed(11)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lr = torch.rand(N, 3, 135, 240, generator=g).cuda()
hb = torch.rand(N, 10, 540, 960, generator=g).cuda()
cnt = [0]
fn0 = F_.conv3x3
def run(*a, **k):
    o = fn0(*a, **k)
    of = o.float() if not k.get("out_nchw_f32") else o.permute(0, 2, 3, 1)
    bad = ~(of.abs() < 1e4)
    n = int(bad.sum())
    if n:
        idx = bad.nonzero()
        bits = (o.view(torch.int16).int() & 0xffff) if o.dtype == torch.bfloat16 else None
        print("call", cnt[0], "in", tuple(a[0].shape), "res" if k.get("res") is not None else "", "srcb" if k.get("src_b") is not None else "",
              "bad", n, "imgs", sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 1].tolist()))[:12],
              "cols%32", sorted(set((idx[:, 2] % 32).tolist())), "cols//32", sorted(set((idx[:, 2] // 32).tolist()))[:40],
              "ch", sorted(set(idx[:, 3].tolist()))[:64], flush=True)
        if bits is not None:
            for kk in range(0, min(idx.shape[0], 64), 8):
                i, r, q, c = idx[kk].tolist()
                print("    img %d row %d col %d ch %d: %04x" % (i, r, q, c, int(bits[i, r, q, c])))
        # rerun the same launch: is it repeatable?
        o2 = fn0(*a, **k)
        print("    rerun equal:", torch.equal(o, o2), "rerun bad:", int((~(o2.float().abs() < 1e4)).sum()))
        o3 = fn0(*a, **k); o4 = fn0(*a, **k)
        print("    rerun3 bad:", int((~(o3.float().abs() < 1e4)).sum()), int((~(o4.float().abs() < 1e4)).sum()))
        sys.exit(0)
    cnt[0] += 1
    return o
F_.conv3x3 = run
for it in range(3):
    cnt[0] = 0
    out = m.infer(lr, hb)
    print("iter", it, "done, out finite", bool(torch.isfinite(out).all()), flush=True)
