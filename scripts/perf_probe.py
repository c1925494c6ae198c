"""Quick per-kernel timing on the GPU box (HIP events on torch's current stream). Prints a table."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd._lib import ACT_GELU

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

dev = "cuda"
print("device", torch.cuda.get_device_name(0))
for dt in (torch.bfloat16, torch.float32):
    for (M, N, K) in [(80640, 2304, 768), (80640, 768, 768), (80640, 3072, 768), (80640, 768, 3072), (2560, 2688, 512)]:
        a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
        ms = timeit(lambda: ops.gemm_nt(a, b))
        print(f"gemm_nt {str(dt)[6:]:8s} {M}x{N}x{K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
    # dW-shaped: [768 x 80640] . [768 x 80640]^T split-K
    for (N, K, M) in [(2304, 768, 80640), (3072, 768, 80640), (768, 3072, 80640)]:
        a = torch.randn(N, M, device=dev).to(dt); b = torch.randn(K, M, device=dev).to(dt)
        out = torch.zeros(N, K, device=dev)
        from idvs.morec_amd.engine import _splitk
        sk = _splitk(N, K, M)
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out, accumulate=2, split_k=sk))
        print(f"gemm dW  {str(dt)[6:]:8s} {N}x{K}x{M} splitk={sk}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
    M, H = 80640, 768
    x = torch.randn(M, 3072, device=dev).to(dt)
    ms = timeit(lambda: ops.transpose(x)); print(f"transpose {str(dt)[6:]} {M}x3072: {ms:.3f} ms {2*x.numel()*x.element_size()/ms/1e6:.0f} GB/s")
    x = torch.randn(M, H, device=dev).to(dt); r = torch.randn(M, H, device=dev).to(dt)
    g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
    ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-12, bias=b, res=r))
    print(f"ln_fwd {str(dt)[6:]} {M}x{H}: {ms:.3f} ms  {4*x.numel()*x.element_size()/ms/1e6:.0f} GB/s")
    y, z, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12, bias=b, res=r)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ms = timeit(lambda: ops.layernorm_bwd(x, r, z, mean, rstd, g, dg, db))
    print(f"ln_bwd {str(dt)[6:]} {M}x{H}: {ms:.3f} ms  {4*x.numel()*x.element_size()/ms/1e6:.0f} GB/s")
    qkv = torch.randn(M, 3 * H, device=dev).to(dt); keep = torch.ones(2688, 30, device=dev)
    desc = ops.attn_desc(2688, 30, 12, 64, False, 0.125, ops.FLT_MIN_MASK, dt)
    ms = timeit(lambda: ops.attn_fwd(desc, qkv, keep)); print(f"attn_fwd {str(dt)[6:]} bert-base: {ms:.3f} ms  {(4*M*H*qkv.element_size())/ms/1e6:.0f} GB/s")
    dctx = torch.randn(M, H, device=dev).to(dt)
    ms = timeit(lambda: ops.attn_bwd(desc, qkv, keep, dctx)); print(f"attn_bwd {str(dt)[6:]} bert-base: {ms:.3f} ms")
