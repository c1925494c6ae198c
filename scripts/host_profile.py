"""Where the HOST time of a launch-bound step goes (cProfile over N fused steps of the BERT-tiny / ID configuration): python scripts/host_profile.py [tiny|id]"""
import cProfile, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from idvs.morec_amd import engine
from idvs.morec_amd.model import BertShape, HipBertModel, Model
from idvs.morec_amd.train_step import TrainStep
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B, S, T, D, item_num, steps = 128, 20, 30, 512, 20000, 60
rng = np.random.default_rng(1)
ids_all = bench.synth_batches(8, B, S, item_num, rng)
counts = np.bincount(ids_all.reshape(-1), minlength=item_num + 1).astype(np.float64) + 1.0
pop = counts / counts[1:].sum(); pop[0] = 1.0
if which == "tiny":
    shape = BertShape.named("tiny")
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2, num_words_title=T, num_words_abstract=50,
                                 num_words_body=50, news_attributes=["title"], bert_model_load="bert_tiny", word_embedding_dim=shape.hidden_size, compute_dtype="fp16")
    m = Model(args, item_num, True, HipBertModel(shape), pop).cuda().train()
    content = bench.synth_catalog(item_num, T, rng)
else:
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2, compute_dtype="fp16")
    m = Model(args, item_num, False, None, pop).cuda().train()
ts = TrainStep(m, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False, defer_update=True)
batches = []
for i in range(8):
    ids = torch.from_numpy(ids_all[i]).cuda()
    if which == "tiny":
        rows = torch.from_numpy(content[ids_all[i].reshape(-1)])
        pack = tuple(t.cuda() for t in engine.token_packing_host(rows[:, T:], rows[:, :T]))
        batches.append((ids.view(-1), rows.cuda(), torch.ones(B, S, device="cuda"), pack))
    else:
        batches.append((ids.view(-1), ids.view(-1).clone(), torch.ones(B, S, device="cuda"), None))
def run(n):
    for i in range(n):
        a, b, c, pk = batches[i % 8]
        ts.step(a, b, c, token_packing=pk)
run(10); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{which}: host issue time {t_host / steps * 1e3:.3f} ms/step, wall {t_all / steps * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable(); run(steps); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
