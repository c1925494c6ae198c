"""Per-parameter gradient error of the mid-size modal model against the CPU oracle for (S, T) pairs: python scripts/longseq_probe.py"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import morec_oracle as orc
from test_model_gpu import make_args, load_det, relerr, DEV
from idvs.morec_amd.model import BertShape, HipBertModel, Model
CASES = [(40, 50, 0), (40, 50, 3), (40, 50, 4)]      # variants 3 / 4: other seeds
for S, T, variant in CASES:
    D, item_num, B = 128, 300, 12
    shape = BertShape(vocab_size=2000, hidden_size=128, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512, max_position_embeddings=64)
    rng = np.random.default_rng({0: 7, 3: 11, 4: 12}.get(variant, 7))
    pop = rng.random(item_num + 1) + 0.05; pop[1:] /= pop[1:].sum(); pop[0] = 1
    args = make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=128, compute_dtype="fp32", num_words_title=T)
    m = load_det(Model(args, item_num, True, HipBertModel(shape), pop)).to(DEV); m.eval()
    content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
    for i in range(1, item_num + 1):
        L = int(rng.integers(3, T + 1)); content[i, :L] = rng.integers(1, 2000, L); content[i, T:T + L] = 1
    ids = np.zeros((B, S + 1), dtype=np.int64); lm = np.zeros((B, S), dtype=np.float32)
    for b in range(B):
        L = int(rng.integers(2, S + 2)); ids[b, S + 1 - L:] = rng.integers(1, item_num + 1, L); lm[b, S + 1 - L:] = 1
    if variant == 1:      # the padding item gets one real token: no zero-length sequence anywhere
        content[0, 0] = 5; content[0, T] = 1
    if variant == 2:      # every user has a full history: no padded behaviour positions
        ids[:] = rng.integers(1, item_num + 1, ids.shape); lm[:] = 1
    items = content[ids.reshape(-1)]
    loss = m(torch.from_numpy(ids).to(DEV).view(-1), torch.from_numpy(items).to(DEV), torch.from_numpy(lm).to(DEV), DEV)
    loss.backward()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ref = orc.model_forward(p, torch.from_numpy(ids).view(-1), torch.from_numpy(items), torch.from_numpy(lm), pop, max_seq_len=S, embedding_dim=D, n_heads=2, use_modal=True, bert_heads=4)
    ref.backward()
    errs = []
    for k, v in m.named_parameters():
        if "pooler" in k or p[k].grad is None or p[k].grad.abs().max().item() < 1e-7: continue
        errs.append((relerr(v.grad.cpu().numpy(), p[k].grad.numpy()), k))
    errs.sort(reverse=True)
    print(f"S={S} T={T} variant {variant}: |dloss| {abs(loss.item() - ref.item()):.2e}; worst: " + "; ".join(f"{k.split('.')[-3:]} {e:.1e}" for e, k in errs[:3]))
    if errs[0][0] > 1e-3:      # how many elements carry the error: a ReLU' flip of ONE pre-activation shows up as one row of w_1 / one entry of its bias
        k = errs[0][1]; d = (dict(m.named_parameters())[k].grad.cpu() - p[k].grad).abs(); thr = 1e-3 * p[k].grad.abs().max()
        print(f"   {k}: {int((d > thr).sum())} of {d.numel()} elements off by more than 1e-3 of the max; rows touched: {sorted(set((d > thr).nonzero()[:, 0].tolist())) if d.dim() == 2 else (d > thr).nonzero().view(-1).tolist()}")
