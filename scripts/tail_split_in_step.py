"""Every ops.gemm_nt call of one bench step, executed with the tail split on and again with it off: per call, the largest
difference of the main output relative to its range.   python scripts/tail_split_in_step.py"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from idvs.morec_amd import ops, _lib
L = _lib.lib()
L.morec_tuning_set(b"gemm8p_tail_split", 1)
real = ops.gemm_nt
seen = {}


def wrapped(x, w, **kw):
    out = real(x, w, **kw)
    if x.dtype != torch.bfloat16 or kw.get("accumulate"):
        return out
    kw2 = dict(kw)
    for k in ("out", "aux_out", "colsum_out"):
        if kw2.get(k) is not None:
            kw2[k] = torch.zeros_like(kw2[k]) if k == "colsum_out" else torch.empty_like(kw2[k])
    L.morec_tuning_set(b"gemm8p_debug", 128)
    ref = real(x, w, **kw2)
    L.morec_tuning_set(b"gemm8p_debug", 0)
    M = kw.get("M") or x.shape[0]; K = kw.get("K") or x.shape[1]; N = kw.get("N") or w.shape[0]
    d = (out.float() - ref.float()).abs().max().item() / (ref.float().abs().max().item() + 1e-30)
    key = (M, N, K, kw.get("act", 0), kw.get("dact", 0))
    if d > seen.get(key, (-1,))[0]:
        seen[key] = (d, int(((out.float() - ref.float()).abs() > 0.02 * ref.float().abs().max()).sum().item()))
    return out


ops.gemm_nt = wrapped
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
try:
    if os.environ.get("IN_STEP_TARGET") == "parity_test":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_bench_mode_parity_gpu as t
        try:
            t.test_bf16_bench_mode_tracks_fp32_parity_mode_at_bench_config()
        except AssertionError as e:
            print("assertion:", e, file=sys.stderr)
    else:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1][0])[:15]:
        print("M,N,K,act,dact =", k, " worst rel diff %.3e, elements off by > 2 %% of range: %d" % v, file=sys.stderr)
