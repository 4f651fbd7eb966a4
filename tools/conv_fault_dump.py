"""Dump the raw bf16 bit patterns of faulty outputs of the 16->64 layer at 4K (diagnosis helper)."""
import sys, torch
sys.path.insert(0, ".")
from superresolution_amd import functional as F_
H, W = 2160, 3840
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(1, 10, H, W, device="cuda", generator=g)
w = torch.randn(64, 10, 3, 3, device="cuda", generator=g) * 0.1
b = torch.zeros(64, device="cuda")
xn = F_.nchw_f32_to_nhwc_bf16(x)
wp = F_.conv3x3_pack_weights(w, 10, xn.shape[-1])
print("ptrs xn %x wp %x b %x" % (xn.data_ptr(), wp.data_ptr(), b.data_ptr()))
for rep in range(3):
    y = F_.conv3x3(xn, wp, b, 64, act=F_.ACT_NONE)
    print("y ptr %x" % y.data_ptr())
    bad = (y.float().abs() > 100)
    idx = bad.nonzero()
    print("rep", rep, "bad", idx.shape[0])
    if idx.shape[0] == 0:
        continue
    bits = y.view(torch.int16).int() & 0xffff
    rows = sorted(set(idx[:, 1].tolist()))[:3]
    for r in rows:
        cols = sorted(set(idx[idx[:, 1] == r][:, 2].tolist()))[:8]
        for c in cols:
            v = bits[0, r, c]
            print("  row %d col %d (px %d of wave %d):" % (r, c, c % 32, (c % 128) // 32),
                  " ".join("%04x" % int(v[ch]) for ch in (32, 33, 34, 40, 41, 42, 48, 49, 56, 57)))
