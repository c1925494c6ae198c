#!/usr/bin/env python
"""What the REFERENCE's own mixed-precision arithmetic does to its own VISION loss (run once in the build container, next to
make_golden_vision.py; the text counterpart is make_autocast_floor.py / g20).

The reference trains the Swin tower under ``torch.cuda.amp.autocast()`` (``V/run.py:210-215``).  This script runs the imported reference vision
``Model`` on the g13 (Swin-T) and g15 (Swin-B) inputs three times -- plain fp32 (= the goldens) and under
``torch.autocast(device_type="cpu", dtype=torch.float16 | torch.bfloat16)``, PyTorch's CPU implementation of the same policy (conv / Linear /
matmul in the low-precision type, LayerNorm / softmax / loss in fp32) -- and writes the losses to ``g21_autocast_floor_vision.json``.  The
|fp32 - autocast| gaps are the reference's own distance to its fp32 loss on these batches: what "the loss of this batch under 16-bit GEMM
operands" can mean, and the yardstick tests/test_swin_gpu.py prints the HIP 16-bit modes against.  Under this policy only stage 1 of a Swin
carries an fp32 residual stream (its input is the fp32 output of the embedding LayerNorm); from the first patch merging on the stream is the
16-bit output of the reduction Linear plus 16-bit sub-layer outputs, i.e. a 16-bit sum -- the data flow of this package's plain 16-bit modes."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_vision as mv  # noqa: E402  (imports the reference vision package)
from transformers import SwinConfig  # noqa: E402


def main():
    out = {}
    for name, tag in (("swin_tiny", "g13"), ("swin_base", "g15")):
        cfg_t = SwinConfig.from_pretrained(f"/root/reference/pretrained_models/{name}").to_dict()
        kw = {k: cfg_t[k] for k in ["image_size", "patch_size", "num_channels", "embed_dim", "depths", "num_heads", "window_size",
                                    "mlp_ratio", "drop_path_rate", "layer_norm_eps"]}
        kw.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        S, D, item_num, B = 3, 256, 12, 2
        args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2, CV_model_load=name)
        pop = np.abs(mv.det_normal(f"pop.{tag}", (item_num + 1,), std=1.0)) + 0.05
        pop = (pop / pop.sum()).astype(np.float32)
        m = mv.RefModel(args, item_num, True, mv.build_swin(kw, D), pop.tolist())
        mv.load_det(m)
        m.eval()
        ids, log_mask = mv.synth_batch(tag, B, S, item_num)
        images = mv.det_normal(f"{tag}.images", (item_num + 1, 3, 224, 224), std=1.0).astype(np.float32)
        images[0] = 0.0
        px = torch.from_numpy(images[ids.reshape(-1)])
        rec = {}
        with torch.no_grad():
            rec["fp32"] = float(m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu"))
            for t, dt in (("autocast_fp16", torch.float16), ("autocast_bf16", torch.bfloat16)):
                try:
                    with torch.autocast(device_type="cpu", dtype=dt):
                        rec[t] = float(m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu"))
                except Exception as e:  # noqa: BLE001
                    rec[t] = None
                    rec[t + "_error"] = f"{type(e).__name__}: {e}"
        rec["rows"] = int(log_mask.sum())
        # gradient norms: fp32 (= the golden's) against the reference under autocast(fp16) with a fixed loss scale of 1024 (what GradScaler does
        # dynamically; tests/test_swin_gpu.py uses the same scale) -- the reference's own gradient-norm gap, parameter by parameter
        def grad_norms(autocast_dt, scale):
            m.zero_grad()
            if autocast_dt is None:
                loss = m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu")
            else:
                with torch.autocast(device_type="cpu", dtype=autocast_dt):
                    loss = m(torch.from_numpy(ids).view(-1), px, torch.from_numpy(log_mask), "cpu")
            (loss * scale).backward()
            return {k: float(p_.grad.double().norm()) / scale for k, p_ in m.named_parameters() if p_.grad is not None}
        g32 = grad_norms(None, 1.0)
        try:
            g16 = grad_norms(torch.float16, 1024.0)
            rel = {k: abs(g16[k] - g32[k]) / (g32[k] + 1e-9) for k in g32 if g32[k] > 1e-6}
            worst = sorted(rel.items(), key=lambda kv: -kv[1])[:5]
            rec["autocast_fp16_grad_norm_relerr_worst"] = float(worst[0][1])
            rec["autocast_fp16_grad_norm_relerr_top5"] = [[k, float(v)] for k, v in worst]
            rec["autocast_fp16_grad_norm_relerr_median"] = float(np.median(list(rel.values())))
        except Exception as e:  # noqa: BLE001
            rec["autocast_fp16_grad_norm_error"] = f"{type(e).__name__}: {e}"
        out[name] = rec
        print(name, rec, flush=True)
    out["note"] = ("losses of the imported reference vision Model on the g13 / g15 inputs: fp32, and under torch.autocast('cpu', fp16 / bf16); torch "
                   + torch.__version__)
    with open(os.path.join(HERE, "g21_autocast_floor_vision.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
