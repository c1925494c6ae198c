"""Latency-class products (SASRec layers, 2 560 rows): gemm_small.hip (forced on every eligible shape) against the 128 x 128 kernel it replaces
(same operands: outputs must be bit-identical) and against torch fp32; times of both.
python scripts/small_gemm_check.py [f16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from idvs.morec_amd import ops, _lib
from idvs.morec_amd._lib import ACT_RELU, DACT_MUL
dev = "cuda"
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
L = _lib.lib()
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
ok = True
for name, M, N, K, kind in [("qkv", 2560, 1536, 512, "nt"), ("o", 2560, 512, 512, "nt"), ("f1+relu", 2560, 2048, 512, "relu"), ("f2", 2560, 512, 2048, "nt"),
                            ("du(dmul+cs)", 2560, 2048, 512, "dmul"), ("dx0", 2560, 512, 1536, "nt"), ("ragged", 2500, 520, 584, "nt"), ("ce dE", 2688, 512, 2560, "nt"),
                            ("qkv D=2048", 640, 6144, 2048, "nt"), ("f2 D=2048", 640, 2048, 8192, "nt")]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(dt); b = (torch.randn(N, K, device=dev) * 0.5).to(dt)
    kw = {}
    if kind == "relu":
        kw = dict(bias=torch.randn(N, device=dev), act=ACT_RELU, aux_out=torch.empty(M, N, device=dev, dtype=dt), aux_deriv=True)
    if kind == "dmul":
        kw = dict(dact=DACT_MUL, dact_in=(torch.rand(M, N, device=dev) > 0.5).to(dt), colsum_out=torch.zeros(N, device=dev))
    outs, us, auxs, css = {}, {}, {}, {}
    for mode in (1, 2):
        L.morec_tuning_set(b"gemm_small", mode)
        if "colsum_out" in kw: kw["colsum_out"].zero_()
        o = ops.gemm_nt(a, b, **kw); torch.cuda.synchronize()
        outs[mode] = o.clone(); auxs[mode] = kw["aux_out"].clone() if "aux_out" in kw else None
        css[mode] = kw["colsum_out"].clone() if "colsum_out" in kw else None
        us[mode] = timeit(lambda: ops.gemm_nt(a, b, **kw))
    same = torch.equal(outs[2], outs[1]) and (auxs[2] is None or torch.equal(auxs[2], auxs[1]))
    ref = a.float() @ b.float().t()
    if kind == "relu": ref = torch.relu(ref + kw["bias"])
    if kind == "dmul": ref = ref * kw["dact_in"].float()
    err = ((outs[2].float() - ref).abs().max() / ref.abs().max()).item()
    cs = "" if css[2] is None else f" colsum relerr {((css[2] - ref.sum(0)).abs().max() / ref.sum(0).abs().max()).item():.1e} vs old {((css[2]-css[1]).abs().max()/css[1].abs().max()).item():.1e}"
    ok &= same and err < 1e-2
    print(f"NT {name:12s} M={M} N={N} K={K}: old {us[1]:6.1f} us  new {us[2]:6.1f} us  ({2.0*M*N*K/us[2]/1e6:6.1f} TF)  bit-identical {same}  relerr vs fp32 {err:.1e}{cs}")
print("OK" if ok else "MISMATCH")
