"""GPU idle time inside the steps of a rocprofv3 kernel trace (rocpd SQLite): per step (a step starts at `marker`, default the text
tower's embedding kernel) the union of the kernels' busy intervals over all streams vs. the step's span, a histogram of the gaps and the
largest ones with the kernels around them.   python scripts/trace_gaps.py kt_results.db [first_step] [marker]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = sys.argv[3] if len(sys.argv) > 3 else "bert_embed_fwd_kernel"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
print(f"{len(rows)} kernels, {len(starts)} steps (marker {marker})")
tot_span = tot_busy = 0.0
agg = {}
big = []
for k in range(first, len(starts) - 1):
    seg = rows[starts[k]:starts[k + 1]]
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy, cs, ce, prev = 0.0, seg[0][1], seg[0][2], seg[0][0]
    for n, s, e in seg[1:]:
        if s > ce:
            busy += ce - cs
            g = (s - ce) / 1e3
            b = "<5us" if g < 5 else "<10us" if g < 10 else "<20us" if g < 20 else "<100us" if g < 100 else ">=100us"
            c, t = agg.get(b, (0, 0.0))
            agg[b] = (c + 1, t + g)
            big.append((g, prev[:60], n[:60]))
            cs, ce = s, e
        else:
            ce = max(ce, e)
        prev = n
    busy += ce - cs
    tot_span += span
    tot_busy += busy / 1e3
n = len(starts) - 1 - first
print(f"steps {first}..{len(starts) - 2}: span {tot_span / n / 1e3:.3f} ms/step, busy {tot_busy / n / 1e3:.3f} ms/step, idle {(tot_span - tot_busy) / n / 1e3:.3f} ms/step")
print("gaps per step:", {k: (round(c / n, 1), round(t / n / 1e3, 3)) for k, (c, t) in sorted(agg.items())}, "(count, ms)")
for g, a, b in sorted(big, reverse=True)[:14]:
    print(f"  {g:8.1f} us  {a:60s} -> {b}")
