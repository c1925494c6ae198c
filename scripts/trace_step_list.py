"""One step of a rocprofv3 kernel trace (rocpd SQLite), kernel by kernel: offset from the step's first kernel, duration, idle gap before
it (vs. the latest end of everything earlier), queue, name -- for the small kernels (< `small_us`) plus their totals, to see what the
torch-side glue costs and where it sits.   python scripts/trace_step_list.py kt_results.db [step] [small_us] [marker]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
small = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
marker = sys.argv[4] if len(sys.argv) > 4 else "bert_embed_fwd_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
# a step's glue runs BEFORE its embedding kernel: cut at the optimizer kernel of the previous step instead
seg = rows[starts[step - 1]:starts[step]]
last_adam = max(i for i, r in enumerate(seg) if "adamw" in r[0])
seg = seg[last_adam + 1:] + rows[starts[step]:starts[step + 1]]
seg = seg[:len(seg) - (len(rows[starts[step]:starts[step + 1]]) - 1 - max(i for i, r in enumerate(rows[starts[step]:starts[step + 1]]) if "adamw" in r[0]))]
t0 = seg[0][1]
hi = seg[0][2]
n_small = t_small = g_small = 0.0
agg = {}
for n, s, e, qid in seg:
    d = (e - s) / 1e3
    gap = max(0.0, (s - hi) / 1e3)
    if d < small:
        n_small += 1; t_small += d; g_small += gap
        key = n.split("(")[0][:70]
        c, t, g = agg.get(key, (0, 0.0, 0.0)); agg[key] = (c + 1, t + d, g + gap)
        print(f"{(s - t0) / 1e3:10.1f} us  {d:7.1f} us  gap {gap:6.1f}  q{qid}  {n[:110]}")
    hi = max(hi, e)
print(f"\nstep {step}: {len(seg)} kernels over {(hi - t0) / 1e6:.3f} ms; kernels shorter than {small} us: {int(n_small)}, {t_small / 1e3:.3f} ms of kernel time, {g_small / 1e3:.3f} ms of idle GPU in front of them")
for k, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"  {c:4d} x {k:70s} {t:8.1f} us  + gaps {g:7.1f} us")
