import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import tests.test_gpu_temporal_train as TT
t = np.load(os.path.join(TT.GOLD, "temporal_train.npz"))
m, args, target, G = TT._model_and_inputs()
m.train()
sr, ew = m(*args)
TT._loss(sr, ew, target, G).backward()
rows = [(rel / TT._bar(name), rel, name, nr, ng / nr) for rel, name, nr, ng in TT._tensor_errors(m, t)]
rows.sort(reverse=True)
for r in rows[:15]:
    print("%.2f rel %.3f %-55s norm %.2e ratio %.3f" % r)
