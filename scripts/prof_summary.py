"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel CSV."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]); title = sys.argv[3] if len(sys.argv) > 3 else ""
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {title}")
print("# kernel, calls, total_ms, avg_us, min_us, max_us, pct")
for r in rows[:45]:
    n = re.sub(r"^void ", "", r[0]); n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)[:80]
    print(f"{n}, {r[1]}, {r[2]/1e6:.3f}, {r[3]/1e3:.1f}, {r[4]/1e3:.1f}, {r[5]/1e3:.1f}, {100*r[2]/tot:.2f}")
print(f"# total kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/steps/1e6:.1f} ms/step")
