"""Which parameters differ first between DEFAULT-mode runs of the five vision fp32 steps of tests/test_deterministic_gpu.py that end on different discrete
branches: the parameter / gradient arenas are copied after every step; for each later run the first step whose parameters differ from run 0 by more
than 1e-6 is printed with the tensors that differ most and the gradient values at those elements.   python scripts/det_probe2.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_deterministic_gpu as t
from idvs.morec_amd import train_step as tsm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tower = sys.argv[2] if len(sys.argv) > 2 else "vision"
snaps = []
orig_step = tsm.TrainStep.step
def step(self, *a, **k):
    out = orig_step(self, *a, **k)
    self.flush(); torch.cuda.synchronize()
    snaps.append(([g["arena"].data.clone() for g in self.groups], [g["arena"].grad.clone() for g in self.groups], [dict(g["arena"].offsets) for g in self.groups]))
    return out
tsm.TrainStep.step = step
runs = []
for i in range(n):
    del snaps[:]
    l, _ = t._run(tower, "fp32", 5, False)
    runs.append((l, list(snaps)))
base = runs[0]
for i, r in enumerate(runs[1:], 1):
    print(f"run {i}: losses {' '.join(f'{v:.7f}' for v in r[0])}   (run 0: {' '.join(f'{v:.7f}' for v in base[0])})")
    for s in range(5):
        worst = []
        for gi, (x, y) in enumerate(zip(r[1][s][0], base[1][s][0])):
            d = (x - y).abs()
            for name, (o, nn, shp) in r[1][s][2][gi].items():
                if nn == 0: continue
                m = float(d[o:o + nn].max())
                if m > 1e-6:
                    j = int(d[o:o + nn].argmax())
                    worst.append((m, name, j, float(r[1][s][1][gi][o + j]), float(base[1][s][1][gi][o + j]), int((d[o:o + nn] > 1e-6).sum()), nn))
        if worst:
            worst.sort(reverse=True)
            print(f"   first difference > 1e-6 after step {s}: {len(worst)} tensors")
            for m, name, j, g1, g0, cnt, nn in worst[:8]:
                print(f"      {name[-64:]:64s} max |dp| {m:.2e} at {j} ({cnt} of {nn} > 1e-6); gradient of that step there {g1:+.3e} / {g0:+.3e}")
            break
    else:
        print("   parameters within 1e-6 after every step")
