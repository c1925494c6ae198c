"""Per-op GPU time of a bench.py run, grouped by (op, tensor shapes): HIP events around every idvs.morec_amd.ops call.

    python scripts/op_profile.py [bench.py arguments ...]      e.g.  --tower swin_tiny --batch 64 --steps 3 --warmup 2

The ops run back to back on one stream, so the event pairs partition the GPU time of a step (nested ops -- an op that
calls other ops -- are listed with their inclusive time and marked)."""
import os, sys, runpy, atexit, collections, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from idvs.morec_amd import ops

LOG = []
DEPTH = [0]


def sig(args, kw):
    out = []
    for x in list(args) + [v for _, v in sorted(kw.items())]:
        if isinstance(x, torch.Tensor):
            out.append("x".join(map(str, x.shape)) + ("f" if x.dtype == torch.float32 else "h" if x.dtype == torch.bfloat16 else "i"))
    return " ".join(out[:5])


def wrap(name, fn):
    def f(*args, **kw):
        if not torch.cuda.is_available():
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d = DEPTH[0]
        DEPTH[0] += 1
        e0.record()
        try:
            return fn(*args, **kw)
        finally:
            e1.record()
            DEPTH[0] -= 1
            extra = ""
            for k in ("act", "dact", "split_m"):
                if kw.get(k):
                    extra += f" {k}={kw[k]}"
            LOG.append((name + extra, sig(args, kw), d, e0, e1))
    return f


for n, fn in list(vars(ops).items()):
    if inspect.isfunction(fn) and fn.__module__ == ops.__name__ and not n.startswith("_") and n not in ("check", "code", "attn_desc", "ce_desc", "swin_attn_desc"):
        setattr(ops, n, wrap(n, fn))


def report():
    if not LOG:
        return
    torch.cuda.synchronize()
    steps = int(os.environ.get("OP_PROFILE_STEPS", "1"))
    agg = collections.OrderedDict()
    for name, s, d, e0, e1 in LOG:
        k = (name, s, d)
        t = e0.elapsed_time(e1)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += t
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for (k, v) in agg.items() if k[2] == 0)
    print(f"# op profile: {len(LOG)} calls, {tot:.1f} ms of top-level op time over the whole run (warm-up included)", file=sys.stderr)
    for (name, s, d), (n, t) in rows[: int(os.environ.get("OP_PROFILE_ROWS", "70"))]:
        print(f"{t:9.2f} ms {100 * t / tot:5.1f}% n={n:5d} avg {1e3 * t / n:8.1f} us  {'  (nested)' if d else ''}{name:28s} {s}", file=sys.stderr)


atexit.register(report)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
