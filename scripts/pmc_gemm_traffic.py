"""FETCH_SIZE / WRITE_SIZE of every GEMM launch from two rocprofv3 --pmc passes -> profiles/<tag>_gemm_pmc.json."""
import json, sqlite3, sys
fetch_db, write_db, out = sys.argv[1:4]
def avg(dbp, counter):
    db = sqlite3.connect(dbp)
    r = db.execute("select sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? and (kernel_name like '%gemm_nt_kernel%' or kernel_name like '%gemm_tn_kernel%')", (counter,)).fetchone()
    return r[0] / r[1], r[1]
f, n = avg(fetch_db, "FETCH_SIZE")
w, _ = avg(write_db, "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (1 x MI355X)",
       "kernels": "gemm_nt_kernel<*> + gemm_tn_kernel, all launches of 3 steps", "launches": n,
       "fetch_size_kb_avg_raw": round(f, 1), "write_size_kb_avg": round(w, 1),
       "correction": "FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated)",
       "hbm_bytes_per_launch_avg": int((2 * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(res)
