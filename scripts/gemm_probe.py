import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
dev = "cuda"
dt = torch.bfloat16
shapes = [(80640, 768, 768), (80640, 2304, 768), (80640, 768, 3072)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    for _ in range(3):
        ops.gemm_nt(a, b)
torch.cuda.synchronize()
