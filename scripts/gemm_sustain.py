"""Why is a GEMM launch slower inside the training step than in a 10-launch microbenchmark?  Same launch (fc1 + GELU + act' out,
M = 51200) timed (a) 10 launches on one buffer set, (b) 300 launches on one buffer set (sustained clocks), (c) 300 launches cycling through
12 buffer sets (the step's memory footprint).   python scripts/gemm_sustain.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops
from idvs.morec_amd._lib import ACT_GELU

dev, dt = "cuda", torch.bfloat16
M, N, K = 51200, 3072, 768
b = torch.randn(N, K, device=dev).to(dt)
bias = torch.zeros(N, device=dev)
sets = [(torch.randn(M, K, device=dev).to(dt), torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt)) for _ in range(12)]


def run(n, nsets):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        a, out, aux = sets[i % nsets]
        ops.gemm_nt(a, b, out=out, bias=bias, act=ACT_GELU, aux_out=aux, aux_deriv=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


run(3, 1)
print(f"10 launches, 1 buffer set  : {run(10, 1):7.1f} us")
print(f"300 launches, 1 buffer set : {run(300, 1):7.1f} us")
print(f"300 launches, 12 buffer sets: {run(300, 12):7.1f} us")
print(f"10 launches, 1 buffer set  : {run(10, 1):7.1f} us (again, right after the sustained runs)")
import time; time.sleep(2.0)
print(f"10 launches, 1 buffer set  : {run(10, 1):7.1f} us (after 2 s idle)")
