"""Per-tile main-loop time of the eight-phase GEMM against K at fixed N: separates the per-K-tile cost from the per-tile constant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from idvs.morec_amd import ops, _lib
dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
L.morec_tuning_set(b"gemm8p", 2)
L.morec_tuning_set(b"gemm8p_debug", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for N in (768, 2304, 3072):
    for K in (768, 3072):
        a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        st = torch.zeros(tiles * 16, device=dev, dtype=torch.int64)
        ops.gemm_nt(a, b, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm_nt(a, b, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        addr = st.data_ptr()
        L.morec_tuning_set(b"gemm8p_stamps_lo", int(np.uint32(addr & 0xffffffff).astype(np.int32)))
        L.morec_tuning_set(b"gemm8p_stamps_hi", int(np.uint32(addr >> 32).astype(np.int32)))
        ops.gemm_nt(a, b, out=out); torch.cuda.synchronize()
        L.morec_tuning_set(b"gemm8p_stamps_lo", 0); L.morec_tuning_set(b"gemm8p_stamps_hi", 0)
        s = st.cpu().numpy().reshape(tiles, 16).astype(np.int64)
        ml = (s[:, 1] - s[:, 0]).astype(np.float64); ep = (s[:, 7] - s[:, 1]).astype(np.float64)
        nk = K // 64
        fine = np.diff(s[:, [0, 8, 9, 10, 11, 12, 13, 14, 15, 1]], axis=1).astype(np.float64).mean(0)
        print("      ctx+zero %5.0f | first wait %5.0f | bars %5.0f | K-tile 0 %5.0f | K-tiles 1-2 %5.0f | rest of steady %6.0f | last two %5.0f | final bar %5.0f | to stamp1 %4.0f" % tuple(fine))
        print(f"N={N:5d} K={K:5d} nk={nk:3d} tiles {tiles:5d}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s | mainloop mean {ml.mean():8.0f} ({ml.mean()/nk:6.0f}/K-tile) "
              f"p10 {np.percentile(ml,10):8.0f} p90 {np.percentile(ml,90):8.0f} | epilogue {ep.mean():6.0f}", flush=True)
