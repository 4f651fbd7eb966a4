"""Per-tensor gradient error of the temporal training graph against tests/golden/temporal_train.npz (dev helper, run on the GPU box)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_temporal_train as T
t = np.load(os.path.join(T.GOLD, "temporal_train.npz"))
m, args, target, G = T._model_and_inputs(); m.train()
sr, ew = m(*args)
loss = T._loss(sr, ew, target, G); loss.backward()
params = dict(m.named_parameters())
rows = []
for i, name in enumerate([str(n) for n in t["names"]]):
    gd = params[name].grad.detach().double().reshape(-1).cpu()
    norm_ref = float(t["norm." + name][0]); gen = torch.Generator().manual_seed(1000 + i); e2 = 0.0
    for k in range(8):
        v = torch.randint(0, 2, (gd.numel(),), generator=gen).double() * 2.0 - 1.0
        e2 += (float((gd * v).sum()) - float(t["proj." + name][k])) ** 2
    rel = (e2 / 8.0) ** 0.5 / max(norm_ref, 1e-12)
    rel2 = None
    if "grad." + name in t.files:
        ref = torch.from_numpy(t["grad." + name]).double().reshape(-1)
        rel2 = float((gd - ref).norm()) / max(norm_ref, 1e-12)
    rows.append((rel, rel2, name, norm_ref, float(gd.norm())))
for r in sorted(rows, reverse=True): print("%.4f %s %-60s ref %.3e gpu %.3e" % (r[0], ("%.4f" % r[1]) if r[1] is not None else "  -   ", r[2], r[3], r[4]))
