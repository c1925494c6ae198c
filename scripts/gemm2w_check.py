"""gemm2w.hip (256 x 128 tiles, two workgroups per CU) against a torch fp32 reference and against gemm8p.hip on the step's shapes.
   python scripts/gemm2w_check.py [check|time|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import _lib, ops  # noqa: E402
from idvs.morec_amd._lib import ACT_GELU  # noqa: E402

DACT_MUL = 3
what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda"


def mode(v):
    _lib.check(_lib.lib().morec_tuning_set(b"gemm2w", v), "tuning")


def run(kind, a, b, bias, dact_in, colsum):
    M, N = a.shape[0], b.shape[0]
    out = torch.empty(M, N, device=dev, dtype=a.dtype)
    if kind == "plain":
        ops.gemm_nt(a, b, bias=bias, out=out)
        return out, None
    if kind == "gelu":
        aux = torch.empty_like(out)
        ops.gemm_nt(a, b, bias=bias, out=out, act=ACT_GELU, aux_out=aux, aux_deriv=True)
        return out, aux
    if kind == "gelu1":
        ops.gemm_nt(a, b, bias=bias, out=out, act=ACT_GELU)
        return out, None
    if kind == "dmul":
        cs = torch.zeros(N, device=dev) if colsum else None
        ops.gemm_nt(a, b, out=out, dact=DACT_MUL, dact_in=dact_in, colsum_out=cs)
        return out, cs
    raise ValueError(kind)


def check():
    worst = 0.0
    for dt in (torch.float16, torch.bfloat16):
        for (M, N, K) in ((54919, 3072, 768), (4096 + 37, 1536, 384), (300, 256, 128), (2048, 136, 64), (5000, 768, 3072)):
            g = torch.Generator(device=dev).manual_seed(M + N + K)
            a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(dt)
            b = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(dt)
            bias = torch.randn(N, device=dev, generator=g) * 0.3
            u = torch.randn(M, N, device=dev, generator=g).to(dt)
            ref_pre = a.float() @ b.float().t()
            for kind in ("plain", "gelu", "gelu1", "dmul", "dmul_cs"):
                mode(2)
                o, x = run(kind.replace("_cs", ""), a, b, bias if "dmul" not in kind else None, u, kind == "dmul_cs")
                mode(1)
                o8, x8 = run(kind.replace("_cs", ""), a, b, bias if "dmul" not in kind else None, u, kind == "dmul_cs")
                torch.cuda.synchronize()
                if kind == "plain":
                    ref = ref_pre + bias
                elif kind in ("gelu", "gelu1"):
                    ref = torch.nn.functional.gelu(ref_pre + bias)
                else:
                    ref = ref_pre * u.float()
                tol = (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * 2
                err = float(((o.float() - ref).abs() / (ref.abs() + 1.0)).max())
                same = bool(torch.equal(o, o8))
                msg = f"{str(dt)[6:]:9s} {M:6d}x{N:5d}x{K:5d} {kind:8s} rel err {err:.2e} bit-equal to gemm8p/generic: {same}"
                if kind == "gelu":
                    pre = (ref_pre + bias).double()
                    dref = (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-pre * pre / 2) / (2 * 3.141592653589793) ** 0.5).float()
                    e2 = float((x.float() - dref).abs().max())
                    msg += f"; act' max abs err {e2:.2e}, equal {bool(torch.equal(x, x8))}"
                    assert e2 < 4 * tol, msg
                if kind == "dmul_cs":
                    e3 = float((x - o.float().sum(0)).abs().max() / (o.float().sum(0).abs().max() + 1e-6))
                    msg += f"; colsum rel err {e3:.2e}"
                    assert e3 < 1e-3, msg
                print(msg, flush=True)
                assert err < tol, msg
                worst = max(worst, err)
    mode(0)
    print("gemm2w check ok; worst rel err", worst)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def time_all():
    dt = torch.float16
    M = int(os.environ.get("SB_M", "54919"))
    shapes = ((3072, 768, "gelu"), (3072, 768, "gelu1"), (3072, 768, "dmul"), (3072, 768, "plain"), (2304, 768, "plain"), (768, 768, "plain"), (768, 3072, "plain"))
    swin = ((137984, 1536, 384, "gelu"), (137984, 1536, 384, "dmul"), (137984, 384, 1536, "plain"), (137984, 1152, 384, "plain"),
            (137984, 384, 384, "plain"), (137984, 384, 1152, "plain"),
            (34496, 3072, 768, "gelu"), (34496, 3072, 768, "dmul"), (34496, 768, 768, "plain"), (34496, 768, 3072, "plain"), (34496, 2304, 768, "plain"),
            (34496, 768, 2304, "plain"),
            # Swin-B (352 images): stage 3 C = 512 (M = 68992), stage 4 C = 1024 (M = 17248)
            (68992, 2048, 512, "gelu"), (68992, 2048, 512, "dmul"), (68992, 512, 2048, "plain"), (68992, 1536, 512, "plain"), (68992, 512, 512, "plain"),
            (17248, 4096, 1024, "gelu"), (17248, 1024, 4096, "plain"), (17248, 3072, 1024, "plain"), (17248, 1024, 1024, "plain"))
    for spec in [(M,) + s for s in shapes] + list(swin):
        Mx, N, K, kind = spec
        a = (torch.randn(Mx, K, device=dev) * 0.5).to(dt)
        b = (torch.randn(N, K, device=dev) * 0.5).to(dt)
        bias = torch.randn(N, device=dev)
        u = torch.randn(Mx, N, device=dev).to(dt) if kind == "dmul" else None
        if os.environ.get("SB_FILL") == "zero":      # quiet operands: the schedule's own rate, with the power limit out of the way
            a.zero_(); b.zero_(); bias.zero_()
            if u is not None:
                u.zero_()
        res = {}
        for rep in range(2):
            for m in (1, 2):
                mode(m)
                us = timeit(lambda: run(kind, a, b, bias if kind != "dmul" else None, u, kind == "dmul"))
                res.setdefault(m, []).append(us)
        mode(0)
        t8, t2 = min(res[1]), min(res[2])
        print(f"M={Mx} N={N} K={K} {kind:6s}: gemm8p {t8:7.1f} us ({2.0 * Mx * N * K / t8 / 1e6:5.0f} TF)   gemm2w {t2:7.1f} us ({2.0 * Mx * N * K / t2 / 1e6:5.0f} TF)   "
              f"x{t8 / t2:.3f}   runs {res}", flush=True)


if what in ("check", "all"):
    check()
if what in ("time", "all"):
    time_all()
