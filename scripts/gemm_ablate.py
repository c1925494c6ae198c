import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
dev="cuda"; dt=torch.bfloat16
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M,N,K) in [(80640,768,768),(80640,768,1536),(80640,768,3072),(80640,2304,768),(20480,768,768),(65536,768,768)]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
    print(f"dbg={os.environ.get('MOREC_GEMM_DBG','0')} {M}x{N}x{K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
