"""Would micro-batches on separate streams hide the HBM-bound kernels (LayerNorm, attention) and the GEMM tail rounds of one half under the
GEMMs of the other?  Forward of the BERT-base item tower (no grad) over 2 688 titles: one stream, then the same rows as 2 / 4 slices on 2 / 4
streams issued back to back."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from idvs.morec_amd.model import BertShape, HipBertModel, Model

dev = "cuda"
T, D, item_num, n = 30, 512, 8000, 2688
shape = BertShape.named("base")
args = types.SimpleNamespace(max_seq_len=20, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2, num_words_title=T,
                             num_words_abstract=50, num_words_body=50, news_attributes=["title"], bert_model_load="bert_base",
                             word_embedding_dim=shape.hidden_size, compute_dtype="fp16")
pop = np.ones(item_num + 1) / item_num
m = Model(args, item_num, True, HipBertModel(shape), pop).to(dev).eval()
content = torch.from_numpy(bench.synth_catalog(item_num, T, np.random.default_rng(1)))[1:n + 1].to(dev)


def run(k):
    streams = [torch.cuda.Stream() for _ in range(k)] if k > 1 else [torch.cuda.current_stream()]
    sl = [content[i * n // k:(i + 1) * n // k].contiguous() for i in range(k)]
    def once():
        if k == 1:
            return [m.bert_encoder(sl[0])]
        outs = []
        cur = torch.cuda.current_stream()
        for s_, x in zip(streams, sl):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs.append(m.bert_encoder(x))
        for s_ in streams:
            cur.wait_stream(s_)
        return outs
    with torch.no_grad():
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            outs = once()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10, torch.cat(outs, 0)


ref = None
for k in (1, 2, 4, 1, 2):
    ms, out = run(k)
    if ref is None:
        ref = out
    print(f"{k} stream(s): {ms:.3f} ms per {n} titles; max |diff| vs one stream {float((out.float() - ref.float()).abs().max()):.2e}", flush=True)
