"""Does a small side-stream kernel (the stand-in for one step of an RCCL ring: a few workgroups, ~10 us of work) make progress while the
persistent eight-phase GEMM holds a workgroup with 160 KiB of LDS and 512 registers per lane-slot on EVERY CU?  A back-to-back train of
GEMM launches on the main stream (the shapes of one encoder layer's backward), a train of small copies on a second stream, each
bracketed by events: latency of the copies alone, under the GEMMs, and under GEMMs that leave r CUs out of their grid
(``morec_tuning_set("gemm8p_reserve_cus", r)``).  python scripts/rccl_cu_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib

dev, dt = "cuda", torch.bfloat16
L = _lib.lib()
M = 54919
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
A = {k: torch.randn(M, k, device=dev).to(dt) for k in (768, 3072)}
W = {(n, k): torch.randn(n, k, device=dev).to(dt) for n, k in shapes}
O = {n: torch.empty(M, n, device=dev, dtype=dt) for n in (768, 3072, 2304)}
side = torch.cuda.Stream()
src = torch.randn(64 * 1024, device=dev)          # 256 KiB copy: a handful of workgroups
dst = torch.empty_like(src)


def gemm_train(reps):
    for _ in range(reps):
        for n, k in shapes:
            ops.gemm_nt(A[k], W[(n, k)], out=O[n])


def run(with_gemm, reserve, n_small=200):
    L.morec_tuning_set(b"gemm8p_reserve_cus", reserve)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = []
    e0.record()
    if with_gemm:
        gemm_train(12)
    e1.record()
    with torch.cuda.stream(side):
        for _ in range(n_small):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            dst.copy_(src, non_blocking=True)
            b.record(side)
            evs.append((a, b))
    torch.cuda.synchronize()
    lat = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    total = evs[0][0].elapsed_time(evs[-1][1]) * 1e3
    return e0.elapsed_time(e1) * 1e3, lat[len(lat) // 2], lat[int(len(lat) * 0.95)], lat[-1], total


gemm_train(2)
run(False, 0)
print("# side-stream copies of 256 KiB (200 in a row) against 48 eight-phase GEMM launches on the main stream")
print("# case                          gemm train us   copy median us   p95 us    max us   copy train us")
for label, wg, r in (("copies alone", False, 0), ("under GEMMs, grid = all CUs", True, 0), ("under GEMMs, 8 CUs reserved", True, 8),
                     ("under GEMMs, 16 CUs reserved", True, 16), ("under GEMMs, 32 CUs reserved", True, 32), ("GEMMs alone, 16 reserved", True, 16)):
    g, med, p95, mx, tot = run(wg, r, n_small=(200 if label != "GEMMs alone, 16 reserved" else 1))
    print(f"{label:30s} {g:12.1f} {med:15.1f} {p95:9.1f} {mx:9.1f} {tot:14.1f}")
L.morec_tuning_set(b"gemm8p_reserve_cus", 0)
