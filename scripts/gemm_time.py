import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from idvs.morec_amd import ops
dev="cuda"; dt=torch.bfloat16
for (M,N,K) in [(80640,768,768),(80640,768,3072),(80640,2304,768)]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    nt = ((M+255)//256)*((N+255)//256)
    tb = torch.zeros(nt*3, device=dev, dtype=torch.int64)
    for _ in range(3): ops.gemm_nt(a, b, out=out, aux_out=tb)
    torch.cuda.synchronize()
    t = tb.cpu().numpy().reshape(-1,3).astype(np.float64)
    main = (t[:,1]-t[:,0]); epi = (t[:,2]-t[:,1]); span = t[:,2].max()-t[:,0].min()
    order = np.argsort(t[:,0])
    print(f"{M}x{N}x{K}: tiles {nt}; wall_clock ticks: mainloop mean {main.mean():.0f} (min {main.min():.0f} max {main.max():.0f}); epilogue mean {epi.mean():.0f} (min {epi.min():.0f} max {epi.max():.0f}); kernel span {span:.0f}; start spread first256 {t[order[:256],0].max()-t[order[0],0]:.0f}; nk={K//64}")
