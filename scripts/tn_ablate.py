import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd.engine import _splitk
dev="cuda"; dt=torch.bfloat16
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
M = 80640
for (N,K) in [(2304,768),(768,768),(3072,768),(768,3072)]:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
    out = torch.zeros(N, K, device=dev)
    for sp in (_splitk(N,K,M), 2*_splitk(N,K,M)):
        ms = timeit(lambda: ops.gemm_tn_(dy, x, out, split_m=sp))
        print(f"dbg={os.environ.get('MOREC_GEMM_DBG','0')} tn {N}x{K}x{M} split={sp}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
