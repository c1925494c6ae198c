"""Time the BERT-base GEMM shapes of the headline step in isolation (bf16): TFLOP/s per shape.
`--lib` adds the vendor library on the same operands (torch.matmul -> hipBLASLt / rocBLAS, plain product only) as a yardstick for the
hand-written kernels: measurement aid only, the product path never calls it.  `python scripts/gemm_bench.py [M] [--lib]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd._lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
LIB = "--lib" in sys.argv
args_ = [x for x in sys.argv[1:] if not x.startswith("--")]
M = int(args_[0]) if args_ else 80640
for name, N, K, kind in [("qkv", 2304, 768, "nt"), ("o", 768, 768, "nt"), ("fc1+gelu", 3072, 768, "gelu"), ("fc2", 768, 3072, "nt"),
                         ("d_fc2(dact)", 3072, 768, "dact"), ("d_qkv", 768, 2304, "nt"), ("w_qkv", 2304, 768, "tn"), ("w_fc2", 768, 3072, "tn")]:
    if kind == "tn":
        from idvs.morec_amd.engine import _splitk
        dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
        out = torch.zeros(N, K, device=dev)
        sp = _splitk(N, K, M)
        us = timeit(lambda: ops.gemm_tn_(dy, x, out, split_m=sp))
    else:
        a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        kw = {}
        if kind == "gelu":
            kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
        if kind == "dact":
            kw = dict(dact=ACT_GELU, dact_in=torch.randn(M, N, device=dev).to(dt))
        us = timeit(lambda: ops.gemm_nt(a, b, out=out, **kw))
    extra = ""
    if LIB:     # same product on the vendor library (no fused epilogue): NT form a @ b.T, TN form dy.T @ x with fp32 output
        if kind == "tn":
            ul = timeit(lambda: torch.matmul(dy.t(), x))
        else:
            ul = timeit(lambda: torch.matmul(a, b.t()))
        extra = f"   | library: {ul:8.1f} us  {2.0 * M * N * K / ul / 1e6:7.1f} TF/s"
    print(f"{name:12s} M={M} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s{extra}")
