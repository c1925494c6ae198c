"""Which amdgpu hwmon belongs to the GPU this process runs on: every card's PCI address / power / shader clock, the torch device's PCI address, and
bench.PowerSampler (matched by PCI address) around 3 s of FFN-up GEMM launches on random and on zero operands."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from idvs.morec_amd import ops
dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(dev)
print("torch device:", pr.name, {k: getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")}, "->", bench._pci_bdf(dev))
for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    rd = lambda n: (open(os.path.join(d, n)).read().strip() if os.path.exists(os.path.join(d, n)) else None)
    print(d, os.path.basename(os.path.realpath(os.path.join(d, "..", ".."))), "power", rd("power1_average") or rd("power1_input"), "sclk", rd("freq1_input"))
M, N, K = 54919, 3072, 768
for label, mk in (("randn", lambda *s: (torch.randn(*s, device=dev) * 0.5).half()), ("zero", lambda *s: torch.zeros(*s, device=dev, dtype=torch.float16))):
    a, b = mk(M, K), mk(N, K)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(5): ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    with bench.PowerSampler(pci_bdf=bench._pci_bdf(dev)) as ps:
        while time.perf_counter() - t0 < 3.0:
            for _ in range(50): ops.gemm_nt(a, b, out=out)
            torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    print(f"{label}: {dt / n * 1e6:.1f} us per launch = {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s;", ps.summary())
