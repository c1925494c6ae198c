"""Per-workgroup s_memtime stamps of the eight-phase GEMM (wave 0): where a tile's time goes.  python scripts/gemm8p_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_GELU
dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
DBG = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L.morec_tuning_set(b"gemm8p_debug", DBG)
L.morec_tuning_set(b"gemm8p", 2)
for name, N, K, kind in [("qkv", 2304, 768, "bias"), ("fc1+gelu", 3072, 768, "gelu"), ("fc2", 768, 3072, "plain"), ("d_fc2(dact+cs)", 3072, 768, "dact")]:
    a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    kw = {}
    if kind == "bias": kw = dict(bias=torch.zeros(N, device=dev))
    if kind == "gelu": kw = dict(bias=torch.zeros(N, device=dev), act=ACT_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
    if kind == "dact": kw = dict(dact=ACT_GELU, dact_in=torch.randn(M, N, device=dev).to(dt), colsum_out=torch.zeros(N, device=dev))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros(tiles * 16, device=dev, dtype=torch.int64)
    ops.gemm_nt(a, b, out=out, **kw); torch.cuda.synchronize()
    addr = st.data_ptr()
    L.morec_tuning_set(b"gemm8p_stamps_lo", int(np.uint32(addr & 0xffffffff).astype(np.int32)))
    L.morec_tuning_set(b"gemm8p_stamps_hi", int(np.uint32(addr >> 32).astype(np.int32)))
    ops.gemm_nt(a, b, out=out, **kw); torch.cuda.synchronize()
    L.morec_tuning_set(b"gemm8p_stamps_lo", 0); L.morec_tuning_set(b"gemm8p_stamps_hi", 0)
    s = st.cpu().numpy().reshape(tiles, 16).astype(np.int64)
    d = np.diff(s[:, :8], axis=1).astype(np.float64)
    t0 = s[:, 0].min()
    names = ["mainloop", "bias", "blk0", "blk1", "blk2", "blk3", "drain"]
    def tm():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm_nt(a, b, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3
    print(f"   debug {DBG}: {min(tm() for _ in range(3)):.1f} us per launch")
    print(f"{name} tiles {tiles}: kernel span {(s[:, 7].max() - t0)} ticks; per-tile total mean {(s[:, 7] - s[:, 0]).mean():.0f}")
    print("   mean  : " + "  ".join(f"{n} {v:8.0f}" for n, v in zip(names, d.mean(0))))
    print("   median: " + "  ".join(f"{n} {v:8.0f}" for n, v in zip(names, np.median(d, 0))))
    # inside the main loop (stamps 8..15 of wave 0): wait for the prologue's first pieces | barriers (+ the wave rows' offset) | first
    # K-tile | second | steady K-tiles | last two | trailing barrier; and from the tile's first stamp to the loop's first
    dm = np.diff(s[:, 8:16], axis=1).astype(np.float64)
    mn = ["wait_prologue", "barrier", "ktile0", "ktile1", "steady", "last2", "trail"]
    print("   mainloop mean: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(mn, dm.mean(0))) + f"  | tile start -> loop start {np.mean(s[:, 8] - s[:, 0]):7.0f}  loop end -> epilogue stamp {np.mean(s[:, 1] - s[:, 15]):7.0f}")
    # start-time distribution: how synchronised are the rounds
    starts = np.sort(s[:, 0] - t0)
    print("   start ticks at tile index 0,255,256,511,512,767: " + ", ".join(str(int(starts[i])) for i in (0, 255, 256, 511, 512, 767) if i < tiles))
