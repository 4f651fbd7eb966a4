"""Run every kernel of the 4K inference twice on identical inputs and report launches whose outputs differ."""
import sys, torch
sys.path.insert(0, ".")
from superresolution_amd import SuperSampler, functional as F_

REPEAT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
count = [0]


def twice(name):
    fn = getattr(F_, name)

    def run(*a, **k):
        ref = fn(*a, **k).clone()
        for r in range(REPEAT):
            out = fn(*a, **k)
            d = out != ref
            n = int(d.sum())
            if n or not bool(torch.isfinite(out.float()).all()):
                idx = d.nonzero()
                info = dict(call=count[0], fn=name, shape=tuple(out.shape), ndiff=n,
                            absmax=float(out.float().abs().max()), refmax=float(ref.float().abs().max()))
                if n:
                    info["lo"] = idx.min(0).values.tolist(); info["hi"] = idx.max(0).values.tolist()
                kw = {q: v for q, v in k.items() if not torch.is_tensor(v)}
                print(info, kw, flush=True)
        count[0] += 1
        return ref
    setattr(F_, name, run)


for nm in ("conv3x3", "upsample_grid_bilinear_nhwc", "nchw_f32_to_nhwc_bf16"):
    twice(nm)
torch.manual_seed(42)
m = SuperSampler(3, 10, 1).cuda().eval()
g = torch.Generator().manual_seed(9)
H, W = 2160, 3840
lr = torch.rand(1, 3, H // 4, W // 4, generator=g).cuda()
hb = torch.rand(1, 10, H, W, generator=g).cuda()
out = m.infer(lr, hb)
print("calls", count[0], "out absmax", float(out.abs().max()), flush=True)
