#!/usr/bin/env python
"""What the REFERENCE's own mixed-precision arithmetic does to its own loss (run once in the build container, next to make_golden.py).

The reference trains under ``torch.cuda.amp.autocast()`` (``T/run.py:242``).  This script runs the imported reference ``Model`` on the g6
inputs (BERT-tiny B = 8, BERT-base B = 2; weights from ``det_param``) three times -- plain fp32 (= golden g6), and under
``torch.autocast(device_type="cpu", dtype=torch.float16 | torch.bfloat16)``, PyTorch's CPU implementation of the same autocast policy
(Linear / matmul in the low-precision type, LayerNorm / softmax / loss in fp32) -- and writes the three losses to
``g20_autocast_floor.json``.  The |fp32 - autocast| gaps are the reference's own distance to its fp32 numbers at these batch sizes: the
FLOOR any implementation with 16-bit GEMM operands lives at (tests/test_fp16_mode_gpu.py prints the HIP modes' gaps beside it)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference package)


def main():
    from transformers import BertConfig
    cfg_base = BertConfig.from_pretrained("/root/reference/pretrained_models/bert_base_uncased").to_dict()
    base_kw = {k: cfg_base[k] for k in ["vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                        "intermediate_size", "max_position_embeddings"]}
    tiny_kw = dict(vocab_size=30522, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=512, max_position_embeddings=512)
    out = {}
    for name, kw, B, wdim in [("tiny", tiny_kw, 8, 128), ("base", base_kw, 2, 768)]:
        S, D, T, item_num = 20, 512, 30, 400
        args = mg.make_args(max_seq_len=S, embedding_dim=D, word_embedding_dim=wdim)
        pop = mg.zipf_pop(item_num, f"pop.g6{name}")
        m = mg.build_modal(args, kw, item_num, pop)
        content = mg.synth_titles(f"g6{name}", item_num, T, 30522)
        ids, log_mask = mg.synth_batch(f"g6{name}", B, S, item_num, ragged=True)
        items = torch.from_numpy(content[ids.reshape(-1)])
        rec = {}
        with torch.no_grad():
            rec["fp32"] = float(m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu"))
            for tag, dt in (("autocast_fp16", torch.float16), ("autocast_bf16", torch.bfloat16)):
                try:
                    with torch.autocast(device_type="cpu", dtype=dt):
                        rec[tag] = float(m(torch.from_numpy(ids).view(-1), items, torch.from_numpy(log_mask), "cpu"))
                except Exception as e:  # noqa: BLE001
                    rec[tag] = None
                    rec[tag + "_error"] = f"{type(e).__name__}: {e}"
        rec["rows"] = int(log_mask.sum())
        out[name] = rec
        print(name, rec)
    out["note"] = ("losses of the imported reference Model on the g6 inputs: fp32, and under torch.autocast('cpu', fp16 / bf16); torch "
                   + torch.__version__)
    with open(os.path.join(HERE, "g20_autocast_floor.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
