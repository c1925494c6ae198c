"""The last `n` kernels of one step of a rocprofv3 kernel trace (rocpd SQLite) with offsets, durations and queues: what is left exposed
after the last encoder layer's backward.   python scripts/trace_step_tail.py kt_results.db [step] [n] [marker]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
marker = sys.argv[4] if len(sys.argv) > 4 else "bert_embed_fwd_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[starts[step]:starts[step + 1]]
last_adam = max(i for i, r in enumerate(seg) if "adamw" in r[0])
seg = seg[:last_adam + 1]
t_end = max(r[2] for r in seg)
for name, s, e, qid in seg[-n:]:
    print(f"{(s - t_end) / 1e3:9.1f} us .. {(e - t_end) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  q{qid}  {name[:100]}")
