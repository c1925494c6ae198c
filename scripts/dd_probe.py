import sys, os, types, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from idvs.morec_amd.model import BertShape, HipBertModel, Model
from idvs.morec_amd.utils.detgen import det_param
DEV="cuda"
S, D, T, item_num, B = 6, 64, 30, 40, 6
shape = BertShape.named("micro")
def run(seed, drop, eps):
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=drop, transformer_block=2,
                                 num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                 bert_model_load="bert_micro", word_embedding_dim=64, compute_dtype="fp32")
    rng = np.random.default_rng(3)
    pop = rng.random(item_num + 1) + 0.05; pop[1:] /= pop[1:].sum(); pop[0] = 1
    bert = HipBertModel(shape, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop)
    m = Model(args, item_num, True, bert, pop)
    with torch.no_grad():
        for k, v in m.state_dict().items(): v.copy_(torch.from_numpy(det_param(k, tuple(v.shape))))
    m = m.to(DEV).train(); torch.manual_seed(seed)
    content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
    for i in range(1, item_num + 1):
        L = int(rng.integers(3, T + 1)); content[i, :L] = rng.integers(1, 512, L); content[i, T:T + L] = 1
    ids = rng.integers(1, item_num + 1, (B, S + 1)); ids[0, :3] = 0
    lm = np.ones((B, S), dtype=np.float32); lm[0, :3] = 0
    tid, tit, tlm = torch.from_numpy(ids).to(DEV).view(-1), torch.from_numpy(content[ids.reshape(-1)]).to(DEV), torch.from_numpy(lm).to(DEV)
    def loss_at():
        m._drop_calls = 0
        return m(tid, tit, tlm, DEV)
    l0 = loss_at(); l0.backward()
    params = [p for n, p in m.named_parameters() if p.grad is not None]
    g = torch.Generator().manual_seed(5)
    vs = [torch.randn(p.shape, generator=g).to(DEV) for p in params]
    dot = sum((p.grad.double() * v.double()).sum().item() for p, v in zip(params, vs))
    with torch.no_grad():
        for p, v in zip(params, vs): p.add_(eps * v)
        lp = loss_at().item()
        for p, v in zip(params, vs): p.sub_(2 * eps * v)
        lm_ = loss_at().item()
    print(f"seed {seed} drop {drop} eps {eps}: loss {l0.item():.4f} fd {(lp-lm_)/(2*eps):.5f} analytic {dot:.5f}", flush=True)
for drop in (0.0, 0.1):
    for eps in (2e-3, 5e-4):
        for seed in (1, 2, 3):
            run(seed, drop, eps)
