import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N) in [(29312, 768), (51000, 768), (80640, 768), (2207744, 96), (137984, 384)]:
    dt = torch.bfloat16
    dy = torch.randn(M, N, device="cuda").to(dt); dy2 = torch.randn(M, N, device="cuda").to(dt); z = torch.randn(M, N, device="cuda").to(dt)
    mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda"); gam = torch.ones(N, device="cuda")
    dg, db, dbias = (torch.zeros(N, device="cuda") for _ in range(3))
    us = timeit(lambda: ops.layernorm_bwd(dy, dy2, z, mean, rstd, gam, dg, db, p_in=0.1, seed_in=5, dbias=dbias))
    us2 = timeit(lambda: ops.layernorm_bwd(dy, dy2, z, mean, rstd, gam, None, None, p_in=0.1, seed_in=5))
    byt = M * N * 2 * 5
    print(f"ln_bwd M={M} N={N}: {us:7.1f} us ({byt / us / 1e3:.0f} GB/s) ; without column grads {us2:7.1f} us ({byt / us2 / 1e3:.0f} GB/s)")
