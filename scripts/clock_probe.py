"""Shader clock and socket power while one GEMM shape runs back to back (rocm-smi sampled from a side thread).
python scripts/clock_probe.py [nt|tn|lib]"""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
dev, dt = "cuda", torch.bfloat16
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
_lib.lib().morec_tuning_set(b"gemm8p", 2)
L = 8192
a = torch.randn(4096, L, device=dev).to(dt); b = torch.randn(4096, L, device=dev).to(dt)
out = torch.empty(4096, 4096, device=dev, dtype=dt)
samples, stop = [], False


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = [l.split(":", 2)[-1].strip() for l in r.splitlines() if "sclk" in l or "Power" in l]
        samples.append(" | ".join(s))
        time.sleep(0.05)


fn = {"nt": lambda: ops.gemm_nt(a, b, out=out), "lib": lambda: torch.matmul(a, b.t())}[kind]
fn(); torch.cuda.synchronize()
t = threading.Thread(target=sampler); t.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); n = 0
e0.record()
while time.time() - t0 < 3.0:
    for _ in range(50):
        fn()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; t.join()
us = e0.elapsed_time(e1) / n * 1e3
print(f"{kind}: {us:.1f} us per launch, {2.0 * 4096 * 4096 * L / us / 1e6:.1f} TF/s sustained over 3 s")
for s in samples[:: max(1, len(samples) // 8)]:
    print("   ", s)
