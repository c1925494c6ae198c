"""Three launches of ONE GEMM of the step (for rocprofv3 counter passes): python scripts/gemm_one.py M N K [nt|gelu|dmul|tn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_GELU, DACT_MUL
M, N, K = (int(x) for x in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
dev, dt = "cuda", (torch.float16 if os.environ.get("MOREC_ONE_DTYPE", "fp16") == "fp16" else torch.bfloat16)
if kind == "tn":
    from idvs.morec_amd.engine import _splitk
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt); out = torch.zeros(N, K, device=dev)
    fn = lambda: ops.gemm_tn_(dy, x, out, split_m=_splitk(N, K, M))
else:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt); out = torch.empty(M, N, device=dev, dtype=dt)
    kw = {}
    if kind == "gelu": kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt), aux_deriv=True)
    if kind == "dmul": kw = dict(dact=DACT_MUL, dact_in=torch.randn(M, N, device=dev).to(dt), colsum_out=torch.zeros(N, device=dev))
    if kind == "bias": kw = dict(bias=torch.zeros(N, device=dev))
    fn = lambda: ops.gemm_nt(a, b, out=out, **kw)
for _ in range(3):
    fn()
torch.cuda.synchronize()
