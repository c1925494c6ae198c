import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
