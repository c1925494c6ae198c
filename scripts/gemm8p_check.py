"""gemm8p (256 x 256 eight-phase bf16 GEMM) against the two-buffer kernel and an fp32 torch product: correctness over
shapes / epilogues, then interleaved A/B timing on the BERT-base shapes.   python scripts/gemm8p_check.py [M] [--lib]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_GELU, ACT_RELU, ACT_NONE

dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
L.morec_tuning_set(b"gemm8p_tail_split", 1)     # exercise the (opt-in) K split of the tail round on the shapes that qualify


def mode(m):
    assert L.morec_tuning_set(b"gemm8p", m) == 0


def ref_epilogue(acc, bias, act, dact, din):
    v = acc if bias is None else acc + bias[None, :]
    pre = v
    if act == ACT_GELU:
        v = torch.nn.functional.gelu(v)
    elif act == ACT_RELU:
        v = torch.relu(v)
    if dact == ACT_GELU:
        u = din.float()
        cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
        pdf = torch.exp(-0.5 * u * u) / (2 * 3.141592653589793) ** 0.5
        v = v * (cdf + u * pdf)
    elif dact == ACT_RELU:
        v = v * (din.float() > 0)
    return v, pre


def check_case(M, N, K, kind, out_dtype=dt):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + K)
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(dt)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(dt)
    kw, bias, din, act, dact = {}, None, None, ACT_NONE, ACT_NONE
    if kind in ("bias", "gelu", "relu"):
        bias = torch.randn(N, device=dev, generator=g)
        kw["bias"] = bias
    if kind == "gelu":
        act = ACT_GELU
    if kind == "relu":
        act = ACT_RELU
    if kind in ("dgelu", "drelu", "dgelu_cs", "drelu_cs"):
        din = torch.randn(M, N, device=dev, generator=g).to(dt)
        dact = ACT_GELU if "gelu" in kind else ACT_RELU
    res = {}
    for m in (1, 2):
        mode(m)
        out = torch.full((M, N), 7.0, device=dev, dtype=out_dtype)
        aux = torch.full((M, N), 7.0, device=dev, dtype=out_dtype) if kind in ("gelu", "relu") else None
        cs = torch.zeros(N, device=dev) if kind.endswith("_cs") else None
        ops.gemm_nt(a, b, out=out, act=act, dact=dact, dact_in=din, aux_out=aux, colsum_out=cs, alpha=0.5 if kind == "alpha" else 1.0, **kw)
        res[m] = (out.float(), None if aux is None else aux.float(), cs)
    acc = a.float() @ b.float().t()
    if kind == "alpha":
        acc = acc * 0.5
    want, pre = ref_epilogue(acc, bias, act, dact, din)
    scale = want.abs().max().item() + 1e-6
    e_new = (res[2][0] - want).abs().max().item() / scale
    e_old = (res[1][0] - want).abs().max().item() / scale
    e_x = (res[2][0] - res[1][0]).abs().max().item() / scale
    msg = f"M={M:6d} N={N:5d} K={K:5d} {kind:9s} {str(out_dtype)[6:]:8s} err new {e_new:.2e} old {e_old:.2e} new-old {e_x:.2e}"
    ok = e_new < (1.2e-2 if out_dtype == dt else 1e-4) and e_new <= 2 * e_old + 1e-6
    if res[2][1] is not None:
        e_aux = (res[2][1] - pre).abs().max().item() / (pre.abs().max().item() + 1e-6)
        msg += f" aux {e_aux:.2e}"
        ok = ok and e_aux < 1.2e-2
    if res[2][2] is not None:
        cs_want = res[2][0].sum(0)
        e_cs = (res[2][2] - cs_want).abs().max().item() / (cs_want.abs().max().item() + 1e-6)
        e_cs_old = (res[1][2] - res[1][0].sum(0)).abs().max().item() / (cs_want.abs().max().item() + 1e-6)
        msg += f" colsum {e_cs:.2e} (old {e_cs_old:.2e})"
        ok = ok and e_cs < 2e-3
    print(("ok   " if ok else "FAIL ") + msg, flush=True)
    return ok


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    args_ = [x for x in sys.argv[1:] if not x.startswith("--")]
    LIB = "--lib" in sys.argv
    ok = True
    if "--no-check" not in sys.argv:
        for M, N, K in [(256, 256, 128), (1000, 512, 192), (3001, 768, 256), (4096, 2304, 768), (2500, 768, 3072), (777, 320, 128), (5000, 1152, 384),
                        (25000, 768, 3072), (23000, 1024, 1536),
                        (5000, 384, 96), (3001, 288, 96), (4096, 256, 200), (2500, 768, 72), (1111, 512, 136)]:   # K not a multiple of 64: partial last K-tile   # the last two: > 256 tiles with a tail round that is split along K
            for kind in ("plain", "bias", "alpha", "gelu", "relu", "dgelu", "drelu", "dgelu_cs", "drelu_cs"):
                ok &= check_case(M, N, K, kind)
            ok &= check_case(M, N, K, "plain", torch.float32)
        # repeatability (races show up as run-to-run differences)
        mode(2)
        a = torch.randn(20000, 768, device=dev).to(dt); b = torch.randn(2304, 768, device=dev).to(dt)
        o0 = ops.gemm_nt(a, b)
        nd = 0
        for _ in range(20):
            nd += int((ops.gemm_nt(a, b) != o0).sum().item())
        print(("ok   " if nd == 0 else "FAIL ") + f"repeatability: {nd} differing elements over 20 runs", flush=True)
        ok &= nd == 0
        print("ALL OK" if ok else "SOME FAILED", flush=True)
    for M in ([int(args_[0])] if args_ else [51200, 80640]):
        print(f"--- timing, M = {M} (us, TF/s); old = two-buffer kernel, new = eight-phase")
        for name, N, K, kind in [("qkv", 2304, 768, "bias"), ("o", 768, 768, "plain"), ("fc1+gelu", 3072, 768, "gelu"), ("fc1+gelu+d", 3072, 768, "gelu_d"), ("fc2", 768, 3072, "plain"),
                                 ("d_fc2(dact+cs)", 3072, 768, "dact"), ("d_fc2(dmul+cs)", 3072, 768, "dmul"), ("d_fc1", 768, 3072, "plain"), ("d_qkv", 768, 2304, "plain")]:
            a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            kw = {}
            if kind == "bias":
                kw = dict(bias=torch.zeros(N, device=dev))
            if kind == "gelu":
                kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
            if kind == "gelu_d":
                kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt), aux_deriv=True)
            if kind == "dact":
                kw = dict(dact=ACT_GELU, dact_in=torch.randn(M, N, device=dev).to(dt), colsum_out=torch.zeros(N, device=dev))
            if kind == "dmul":
                kw = dict(dact=_lib.DACT_MUL, dact_in=torch.randn(M, N, device=dev).to(dt), colsum_out=torch.zeros(N, device=dev))
            t = {1: [], 2: []}
            for rnd in range(3):
                for m in (1, 2):
                    mode(m)
                    t[m].append(timeit(lambda: ops.gemm_nt(a, b, out=out, **kw)))
            fl = 2.0 * M * N * K
            extra = ""
            if LIB:
                ul = timeit(lambda: torch.matmul(a, b.t()))
                extra = f" | library {ul:7.1f} us {fl / ul / 1e6:7.1f} TF/s"
            print(f"{name:15s} N={N:5d} K={K:5d}: old {min(t[1]):7.1f} us {fl / min(t[1]) / 1e6:7.1f} | new {min(t[2]):7.1f} us {fl / min(t[2]) / 1e6:7.1f}{extra}", flush=True)
    mode(0)


if __name__ == "__main__":
    main()
