"""The kernels of one step between the encoder's last forward attention and its first backward attention (projection head, the
recommender's forward / backward, scoring): offsets, durations, idle gap before each, queue.   python scripts/trace_step_middle.py kt_results.db [step]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "bert_embed_fwd_kernel" in r[0]]
seg = rows[starts[step]:starts[step + 1]]
af = [i for i, r in enumerate(seg) if "attn_fwd_mfma_kernel" in r[0]]
ab = [i for i, r in enumerate(seg) if "attn_bwd_mfma_kernel" in r[0]]
lo = af[11] if len(af) > 11 else af[-1]
hi_i = [i for i in ab if i > lo][2]      # the first two backward attentions are the recommender's
t0 = seg[lo][1]
hi = seg[lo][1]
idle = 0.0
for n, s, e, qid in seg[lo:hi_i + 1]:
    gap = max(0.0, (s - hi) / 1e3)
    idle += gap
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  q{qid}  {n[:105]}")
    hi = max(hi, e)
print(f"segment: {(hi - t0) / 1e3:.1f} us, {hi_i - lo + 1} kernels, idle {idle:.1f} us")
