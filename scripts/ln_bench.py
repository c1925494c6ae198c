"""LayerNorm forward / backward launch times at the bench's token count (HBM-bound kernels: bytes moved / time).
python scripts/ln_bench.py [M] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(M, N, device=dev).to(dt); res = torch.randn(M, N, device=dev).to(dt)
gm = torch.ones(N, device=dev); bt = torch.zeros(N, device=dev); bias = torch.zeros(N, device=dev)
row = M * N * 2 / 1e6
for p in (0.0, 0.1):
    t = timeit(lambda: ops.layernorm_fwd(x, gm, bt, 1e-12, bias=bias, res=res, p_in=p, seed_in=3))
    print(f"ln_fwd p_in={p}: {t:7.1f} us  {4 * row / t:6.2f} TB/s (x, res -> z, y)")
    y, z, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-12, bias=bias, res=res, p_in=p, seed_in=3)
    da = torch.randn(M, N, device=dev).to(dt); db = torch.randn(M, N, device=dev).to(dt)
    dg = torch.zeros(N, device=dev); dbt = torch.zeros(N, device=dev); dbias = torch.zeros(N, device=dev)
    t = timeit(lambda: ops.layernorm_bwd(da, db, z, mean, rstd, gm, dg, dbt, p_in=p, seed_in=3, dbias=dbias))
    nt = 5 if p > 0 else 4
    print(f"ln_bwd p_in={p}: {t:7.1f} us  {nt * row / t:6.2f} TB/s (dy_a, dy_b, z -> dz{', dzd' if p > 0 else ''})")
