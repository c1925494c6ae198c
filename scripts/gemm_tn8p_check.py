"""Weight-gradient GEMM (dW = dY^T X): the eight-phase kernel against the two-buffer kernel and an fp32 torch product, then A/B
timing on the BERT-base shapes.   python scripts/gemm_tn8p_check.py [M] [--no-check]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd.engine import _splitk
dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
def mode(m): assert L.morec_tuning_set(b"gemm8p", m) == 0
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ok = True
if "--no-check" not in sys.argv:
    for M, N, K, sp in [(1024, 256, 256, 2), (5000, 768, 768, 4), (4133, 2304, 768, 3), (9999, 768, 3072, 7), (3000, 320, 136, 2), (2049, 512, 2048, 2),
                        (700, 256, 512, 1), (40000, 3072, 768, 9)]:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(dt); x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(dt)
        init = torch.randn(N, K, device=dev, generator=g)
        res = {}
        for m in (1, 2):
            mode(m)
            out = init.clone()
            ops.gemm_tn_(dy, x, out, split_m=sp, accumulate=(sp > 1))
            res[m] = out.clone()
        want = dy.float().t() @ x.float() + (init if sp > 1 else 0)
        sc = want.abs().max().item()
        e2, e1, ex = [(a - b).abs().max().item() / sc for a, b in ((res[2], want), (res[1], want), (res[2], res[1]))]
        good = e2 < 2e-5 * (M ** 0.5) / 30 + 2e-6 and e2 <= 3 * e1 + 1e-6
        ok &= good
        print(("ok   " if good else "FAIL ") + f"M={M:6d} N={N:5d} K={K:5d} split {sp}: err new {e2:.2e} old {e1:.2e} new-old {ex:.2e}", flush=True)
    mode(2)
    dy = torch.randn(20000, 768, device=dev).to(dt); x = torch.randn(20000, 768, device=dev).to(dt)
    o0 = torch.zeros(768, 768, device=dev); ops.gemm_tn_(dy, x, o0, split_m=4)
    nd = 0
    for _ in range(10):
        o1 = torch.zeros(768, 768, device=dev); ops.gemm_tn_(dy, x, o1, split_m=4)
        nd += int((o1 != o0).sum().item())
    print(("ok   " if nd == 0 else "FAIL ") + f"repeatability: {nd} differing elements over 10 runs")
    ok &= nd == 0
    print("ALL OK" if ok else "SOME FAILED", flush=True)
M = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 51200
print(f"--- timing, M = {M} (us, TF/s incl. the slab reduction)")
for name, N, K in [("w_qkv", 2304, 768), ("w_o", 768, 768), ("w_fc1", 3072, 768), ("w_fc2", 768, 3072)]:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
    out = torch.zeros(N, K, device=dev)
    sp = _splitk(N, K, M)
    t = {1: [], 2: []}
    for rnd in range(3):
        for m in (1, 2):
            mode(m)
            t[m].append(timeit(lambda: ops.gemm_tn_(dy, x, out, split_m=sp)))
    fl = 2.0 * M * N * K
    print(f"{name:6s} N={N:5d} K={K:5d} split {sp:2d}: old {min(t[1]):7.1f} us {fl / min(t[1]) / 1e6:7.1f} | new {min(t[2]):7.1f} us {fl / min(t[2]) / 1e6:7.1f}", flush=True)
mode(0)
