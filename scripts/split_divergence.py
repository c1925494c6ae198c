"""Where does a bf16 step with the gemm8p tail split ON first diverge from the same step with it OFF?  Every ops.* call of one
forward + backward is checksummed (sum |x| of every tensor it returns) in both runs.   python scripts/split_divergence.py"""
import os, sys, types, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from idvs.morec_amd import ops, _lib
from idvs.morec_amd.model import BertShape
from idvs.morec_amd.train_step import TrainStep
import test_bench_mode_parity_gpu as t

L = _lib.lib()
L.morec_tuning_set(b"gemm8p_tail_split", 1)
B, S, T, D, item_num = 128, 20, 30, 512, 20000
shape = BertShape.named("base")
rng = np.random.default_rng(12345)
content = bench.synth_catalog(item_num, T, rng)
ids_all = bench.synth_batches(2, B, S, item_num, np.random.default_rng(13345))
counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
pop = counts / counts[1:].sum(); pop[0] = 1.0
m0 = t._build("bf16", shape, item_num, pop, S, T, D)
state = {k: v.detach().cpu().clone() for k, v in m0.state_dict().items()}
del m0
log = []
for n, fn in list(vars(ops).items()):
    if inspect.isfunction(fn) and fn.__module__ == ops.__name__ and not n.startswith("_") and n not in ("check", "code", "attn_desc", "ce_desc", "swin_attn_desc"):
        def mk(n, fn):
            def f(*a, **k):
                r = fn(*a, **k)
                outs = r if isinstance(r, (tuple, list)) else (r,)
                cs = [float(x.detach().double().abs().sum()) for x in outs if isinstance(x, torch.Tensor) and x.is_floating_point()]
                for key in ("out", "aux_out", "colsum_out"):
                    if isinstance(k.get(key), torch.Tensor):
                        cs.append(float(k[key].detach().double().abs().sum()))
                shp = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:2]
                log.append((n, shp, cs))
                return r
            return f
        setattr(ops, n, mk(n, fn))
runs = {}
for name, dbg, tb in (("off", 128, 2), ("on", 0, 2), ("on_bias6", 0, 6), ("on_bias10", 0, 10), ("legacy", 128, 2)):
    L.morec_tuning_set(b"gemm8p_debug", dbg)
    L.morec_tuning_set(b"gemm8p_tail_bias", tb)
    L.morec_tuning_set(b"gemm8p", 1 if name == "legacy" else 0)
    m = t._build("bf16", shape, item_num, pop, S, T, D, state)
    ts = TrainStep(m, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)
    del log[:]
    ids = torch.from_numpy(ids_all[0]).cuda(); items = torch.from_numpy(content[ids_all[0].reshape(-1)]).cuda()
    loss = ts.forward_backward(ids.view(-1), items, torch.ones(B, S, device="cuda"))
    torch.cuda.synchronize()
    runs[name] = (float(loss), list(log))
    del ts, m
    torch.cuda.empty_cache()
print({k: round(v[0], 5) for k, v in runs.items()})
shown = 0
for i, (a, b) in enumerate(zip(runs["off"][1], runs["on"][1])):
    if a[0] != b[0] or len(a[2]) != len(b[2]):
        print("call sequence differs at", i, a[:2], b[:2]); break
    rel = max([abs(x - y) / (abs(x) + 1e-30) for x, y in zip(a[2], b[2])] + [0.0])
    if rel > 1e-4 and shown < 12:
        print(f"call {i:4d} {a[0]:20s} {a[1]} checksum rel diff {rel:.2e}")
        shown += 1
