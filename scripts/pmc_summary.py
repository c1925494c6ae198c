"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (one --pmc pass)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else "%"
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
if len(sys.argv) > 3: print(cols)
kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
rows = db.execute(f"select {kn}, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where {kn} like ? group by {kn}, counter_name", (pat,)).fetchall()
for r in rows:
    print(f"{r[0][:60]:60s} {r[1]:36s} {r[2] / max(1, r[3]):16.1f}  (dispatches {r[3]})")
