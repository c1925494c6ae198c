"""HBM traffic of the step's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as the TCC slots
require) of the SAME bench.py command -> profiles/<tag>_gemm_pmc.json, with the signature of the profiled run (GEMM launches and
FLOPs per step) so that bench.py can refuse the file when it does not describe the run it is asked about.
    python scripts/pmc_traffic.py fetch.db write.db bench_line.json out.json"""
import json, re, sqlite3, sys
fetch_db, write_db, bench_json, out = sys.argv[1:5]
line = json.loads([l for l in open(bench_json).read().splitlines() if l.startswith("{")][-1])
steps = line.get("steps_executed") or (line["steps"] + line["warmup"])     # every step the profiled process ran (warm-up, timed, instrumented pass)

def per_kernel(dbp, counter):
    db = sqlite3.connect(dbp)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
    res = {}
    for name, s, n in db.execute(f"select {kn}, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? group by {kn}", (counter,)):
        k = re.sub(r"^void ", "", name); k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"\(.*", "", k)
        res[k] = (s, n)
    return res

f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
is_gemm = lambda k: k.startswith(("gemm8p_kernel", "gemm_tn8p_kernel", "gemm_nt_kernel", "gemm_tn_kernel", "gemm_skinny_kernel", "gemm_skinny_wide_kernel"))
kernels = {}
for k in sorted(set(f) | set(w)):
    fs, fn = f.get(k, (0.0, 0)); ws, wn = w.get(k, (0.0, 0))
    n = max(fn, wn)
    if n == 0: continue
    # FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH doubled on gfx950 (128-byte requests tallied as 64 B: MI355X_MICROARCH.md, HBM)
    kernels[k] = {"launches": n, "fetch_kib_raw_per_launch": round(fs / n, 1), "write_kib_per_launch": round(ws / n, 1),
                  "hbm_bytes_per_launch": int((2 * fs + ws) / n * 1024)}
g_launch = sum(v["launches"] for k, v in kernels.items() if is_gemm(k))
g_bytes = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in kernels.items() if is_gemm(k))
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python bench.py --steps %d --warmup %d --no-cpu-baseline --no-secondary (1 x MI355X)" % (line["steps"], line["warmup"]),
       "correction": "FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported (uncalibrated)",
       "signature": {"gemm_launches_per_step": g_launch / steps, "gemm_flops_per_step": line["roofline"].get("gemm_flops_per_step"),
                     "token_layout": line["config"].get("token_layout"), "workload": line["config"]["workload"]},
       "gemm_kernels": "gemm8p_kernel<*> + gemm_tn8p_kernel<*> + gemm_nt_kernel<*> + gemm_tn_kernel<*> + gemm_skinny_kernel<*> + gemm_skinny_wide_kernel<*>, all launches of %d steps" % steps,
       "launches": g_launch, "hbm_bytes_per_launch_avg": int(g_bytes / max(1, g_launch)), "per_kernel": kernels}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("signature", "launches", "hbm_bytes_per_launch_avg")}))
