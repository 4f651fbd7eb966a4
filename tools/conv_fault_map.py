"""Compare one 3x3 layer against torch's conv2d and print where the results differ.
usage: conv_fault_map.py cin cpad cout n h w [res]"""
import sys, torch
sys.path.insert(0, ".")
from superresolution_amd import functional as F_
cin, cpad, cout, N, H, W = [int(v) for v in sys.argv[1:7]]
use_res = len(sys.argv) > 7
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(N, cin, H, W, device="cuda", generator=g)
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.1
b = torch.randn(cout, device="cuda", generator=g) * 0.1
xn = F_.nchw_f32_to_nhwc_bf16(x)
assert xn.shape[-1] == cpad
res = (torch.rand(N, H, W, cout, device="cuda", generator=g) - 0.5).to(torch.bfloat16) if use_res else None
wp = F_.conv3x3_pack_weights(w, cin, cpad)
ref = torch.relu(torch.nn.functional.conv2d(xn[..., :cin].permute(0, 3, 1, 2).float(), w.bfloat16().float(), b, padding=1))
if use_res:
    ref = ref + res.permute(0, 3, 1, 2).float()
for rep in range(4):
    y = F_.conv3x3(xn, wp, b, cout, res=res, act=F_.ACT_RELU)
    d = (y.permute(0, 3, 1, 2).float() - ref).abs()
    bad = ~(d <= 0.05)
    n = int(bad.sum())
    print("rep", rep, "bad", n, "max err", float(d.max()), flush=True)
    if n:
        idx = bad.nonzero()
        im, ch, rows, cols = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        print("  images", sorted(set(im.tolist())), "channels", sorted(set(ch.tolist()))[:64])
        print("  rows", sorted(set(rows.tolist()))[:40])
        print("  cols%32", sorted(set((cols % 32).tolist())), "cols//32", sorted(set((cols // 32).tolist())))
        bits = y.view(torch.int16).int() & 0xffff
        for k in range(0, min(idx.shape[0], 400), 50):
            i, c, r, q = idx[k].tolist()
            print("   img %d row %d col %d ch %d: got %04x ref %.4f" % (i, r, q, c, int(bits[i, r, q, c]), float(ref[i, c, r, q])))
