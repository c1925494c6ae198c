import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(8192, 8192, 16384), (8192, 8192, 4096), (8192, 8192, 1024), (8192, 8192, 256), (16384, 4096, 768), (65536, 768, 768)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: ops.gemm_nt(a, b, out=out))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"M={M} N={N} K={K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  tiles {tiles} ({tiles / 256:.2f} rounds), {us / (tiles / 256) :.1f} us per round, {K // 64} stages")
