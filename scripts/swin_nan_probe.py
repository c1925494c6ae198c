"""Where does the bf16 Swin-T run of tests/test_bench_mode_parity_vision_gpu.py go non-finite?  python scripts/swin_nan_probe.py [reps]"""
import dataclasses, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from idvs.morec_amd.model import Model
from idvs.morec_amd.model.swin import HipSwinForImageClassification
from idvs.morec_amd.swin_engine import SwinShape
from idvs.morec_amd.train_step import TrainStep
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, S, D, item_num, steps = 16, 10, 2048, 600, 10
shape = dataclasses.replace(SwinShape.named("swin_tiny"), drop_path_rate=0.0)
rng = np.random.default_rng(4321)
ids_all = rng.integers(1, item_num + 1, size=(steps, B, S + 1)).astype(np.int64)
counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
pop = counts / counts[1:].sum(); pop[0] = 1.0
gen = torch.Generator(device="cuda").manual_seed(4321)
catalog = torch.randn((item_num + 1, 3, shape.image_size, shape.image_size), device="cuda", generator=gen); catalog[0].zero_()
args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2, CV_model_load="swin_tiny", compute_dtype="bf16")
n_bad = 0
for r in range(reps):
    torch.manual_seed(12345)
    m = Model(args, item_num, True, HipSwinForImageClassification(shape, D), pop).to("cuda").train()
    ts = TrainStep(m, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)
    losses, bad = [], None
    for i in range(steps):
        ids = torch.from_numpy(ids_all[i]).cuda().view(-1)
        loss = ts.forward_backward(ids, catalog[ids], torch.ones(B, S, device="cuda"))
        losses.append(float(loss))
        if bad is None:
            names = []
            for g in ts.groups:
                a = g["arena"]
                if not bool(torch.isfinite(a.grad).all()):
                    names += [n for n in a.offsets if not bool(torch.isfinite(a.view(a.grad, n)).all())]
            pbad = [n for g in ts.groups for n in g["arena"].offsets if not bool(torch.isfinite(g["arena"].view(g["arena"].data, n)).all())]
            if names or pbad or not np.isfinite(losses[-1]):
                bad = (i, len(names), names[:3], names[-2:], len(pbad), pbad[:3])
        ts.reduce_gradients(); ts.optimizer_step()
    n_bad += bad is not None
    print(f"rep {r}: losses {['%.3f' % x for x in losses]}" + (f"  FIRST NON-FINITE at step {bad[0]}: {bad[1]} grads e.g. {bad[2]} ... {bad[3]}; {bad[4]} params e.g. {bad[5]}" if bad else ""), flush=True)
    del ts, m
    torch.cuda.empty_cache()
print(f"non-finite runs: {n_bad}/{reps}")
