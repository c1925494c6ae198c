"""One step of a rocprofv3 kernel trace (rocpd SQLite) in launch order with grid / workgroup sizes: offset, duration, gap, queue, grid, name.
python scripts/trace_step_full.py kt_results.db <marker kernel substring> [step]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print("# columns:", cols)
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "0")
lds = "lds_size" if "lds_size" in cols else ("lds_block_size" if "lds_block_size" in cols else "0")
rows = db.execute(f"select name, start, end, {q}, {gx}, {wx}, {lds} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[starts[step]:starts[step + 1]]
t0 = seg[0][1]; hi = seg[0][1]; tk = 0.0
for n, s, e, qid, g, w, l in seg:
    d = (e - s) / 1e3; gap = (s - hi) / 1e3; tk += d
    print(f"{(s - t0) / 1e3:9.1f} us {d:7.1f} us gap {gap:6.1f} q{qid} grid {g:>8} wg {w:>4} lds {l:>6} {n[:120]}")
    hi = max(hi, e)
print(f"# step {step}: {len(seg)} kernels, {tk/1e3:.3f} ms of kernel time over {(hi - t0)/1e6:.3f} ms")
