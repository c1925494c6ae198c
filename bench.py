#!/usr/bin/env python
"""bench.py -- end-to-end MoRec in-batch train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic MIND-shaped input, per rank:
H2D of (ids, token rows, log_mask) -> BERT-base item encoder fwd -> SASRec -> fused in-batch debiased CE
-> full backward -> gradient all-reduce (N > 1; negatives pooled over ranks by all-gather) -> fused AdamW.
Workload: BASELINE.json configs[2] = SASRec + BERT-base, B = 128 user sequences per GPU, S = 20 (raw
history 23), 30-token titles, D = 512, 2 heads, 2 blocks; fp16 MFMA operands / fp32 accumulate / fp32 master
weights with the GradScaler protocol on the device (the reference runs fp16 autocast + GradScaler, T/run.py:210,242-247).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="user sequences per GPU")
    ap.add_argument("--bert", default="base")
    ap.add_argument("--tower", default="text", help="text (BASELINE.json metric: BERT item encoder) | swin_tiny | swin_base | swin_micro "
                    "(vision configs of BASELINE.json: Swin item encoder, S=10, D=2048, 224x224 images; default --batch 64)")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32", "fp32x3", "fp16_res32", "bf16_res32"],
                    help="fp16 (default): IEEE-half operands on the MFMA + GradScaler loss scaling -- the reference's own GPU arithmetic "
                         "(T/run.py:210,242-247; V/run.py likewise) and the 16-bit mode whose loss stays within north_star's 1e-3 of the fp32 parity mode; bf16: the "
                         "same kernels on bf16 operands (no loss scaling); fp32 / fp32x3: the parity modes; fp16_res32 / bf16_res32: 16-bit GEMMs with an "
                         "fp32 residual stream (LayerNorm in and out fp32) -- the data flow of torch.cuda.amp.autocast itself")
    ap.add_argument("--vision-input", default="resident", choices=["resident", "u8"],
                    help="vision towers: resident = fp32 NCHW catalogue in HBM, a step gathers its images on the device (what V/run.py:201-204 has after "
                         "the DataLoader); u8 = the input pipeline inside the timed step (SURVEY §8 a3 / f3): decoded uint8 images of --native-size on the "
                         "HOST, packed by a collate thread, uploaded, resampled on the GPU (morec_image_resize_u8, Pillow-exact) and normalised in the patch "
                         "im2col (morec_swin_patchify_u8)")
    ap.add_argument("--native-size", type=int, default=256, help="--vision-input u8: side of the synthetic decoded images before the resize")
    ap.add_argument("--item-num", type=int, default=80000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--dedup", action="store_true", help="encode each distinct item of the batch once (SURVEY §8(f)-2; opt-in: with "
                    "dropout on, duplicates then share a mask). The default run reports it as a secondary measurement only.")
    ap.add_argument("--padded", action="store_true", help="text tower: run the encoder layers on all T positions of every title (the "
                    "reference's shape of compute) instead of the real tokens only; the default run reports it as a secondary line")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (item-dedup) measurement after the timed region")
    ap.add_argument("--no-pool", action="store_true", help="rank-local negatives (reference behaviour) instead of the pooled set")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (functional smoke test of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--graph", action="store_true", help="one rank: replay the step as a captured hipGraph (TrainStep.step_graphed; one graph per input "
                    "shape, the unpadded token layout padded to buckets of 512 rows) instead of launching its ~700 kernels from the host.  Opt-in: on "
                    "this stack a replayed graph keeps the 5-8 us dependency gap between consecutive kernels, so it buys little (DESIGN.md)")
    ap.add_argument("--no-sweep", action="store_true", help="N > 1: skip the reduced default sweep (torch.distributed x reserved CUs x overlap)")
    ap.add_argument("--sweep", action="store_true", help="N > 1: after the headline region, time the data-parallel knobs -- collectives through "
                    "torch.distributed vs the library's own RCCL communicators, 0 / 8 / 16 CUs reserved for the ring kernel, reduction overlapped "
                    "with the backward or after it -- and print each configuration's bucket trace under `sweep` (schema: INTEGRATION.md)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only: every rank joins the process group, rank 0 prints the "
                    "world size it observed as one JSON line and exits (no GPU work; used by the CPU test of the N > 1 launch path)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-timeout", type=int, default=150)
    a = ap.parse_args()
    a.dtype16 = a.dtype.split("_")[0]      # storage type of the GEMM operands ("fp16_res32" -> "fp16")
    return a


_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def synth_catalog(item_num, T, rng):
    """item_content int64 [item_num + 1, 2T]: [CLS] tokens [SEP] PAD..., mask; row 0 all zero."""
    content = np.zeros((item_num + 1, 2 * T), dtype=np.int64)
    lens = rng.integers(8, T + 1, item_num)
    toks = rng.integers(1000, 30522, (item_num, T))
    ar = np.arange(T)[None, :]
    valid = ar < lens[:, None]
    toks = np.where(valid, toks, 0)
    toks[:, 0] = 101
    toks[np.arange(item_num), lens - 1] = 102
    content[1:, :T] = toks
    content[1:, T:] = valid
    return content


def synth_batches(n, B, S, item_num, rng):
    """Full-length train sequences (raw history 23 -> S + 1 = 21 items, no padding), Zipf(1.0) popularity."""
    w = 1.0 / np.arange(1, item_num + 1)
    w /= w.sum()
    perm = rng.permutation(item_num) + 1
    ids = perm[rng.choice(item_num, size=(n, B, S + 1), p=w)]
    return ids.astype(np.int64)


def self_launch(a):
    """``python bench.py --gpus N`` outside a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, the way
    the reference's launcher does (T/train_bert_base.py:40-50: ``torch.distributed.launch --nproc_per_node N run.py ...``), so
    that an N-GPU request can never silently run on one GPU.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {a.gpus} without a launcher environment: starting {a.gpus} ranks ({' '.join(cmd[1:9])} ...)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


class PowerSampler:
    """Socket power and shader clock of GPU 0 from the amdgpu hwmon files (readable by an ordinary user), sampled every `period` s on a side
    thread while the `sustained` pass runs: the evidence behind DESIGN.md's "power-limited" reading of the GEMM roofline (a 16-bit MFMA
    main loop draws the part's whole budget below its 2.4 GHz data-sheet clock, which is the clock the 2.5 PFLOP/s peak assumes)."""

    def __init__(self, period=0.02, pci_bdf=None):
        """`pci_bdf` ("0000:bb:dd.f"): the PCI address of the GPU the step runs on.  A one-GPU lease on a multi-GPU host still sees every card's hwmon
        files; without the match the first card with a power file would be sampled -- a neighbour's load (seen in round 6: 342 W / 948 W / 1352 W on
        three boxes for the same step).  `matched` says whether the sampled card is the step's own; unmatched samples are reported, not interpreted."""
        import glob
        self.period, self.rows, self._stop, self._thr = period, [], False, None
        self.files, self.matched = {}, False
        first = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = {k: os.path.join(d, n) for k, names in (("power", ("power1_average", "power1_input")), ("cap", ("power1_cap",)),
                                                         ("sclk", ("freq1_input",))) for n in names if os.path.exists(os.path.join(d, n))}
            if "power" not in f:
                continue
            if first is None:
                first = f
            bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
            if pci_bdf and bdf.lower() == pci_bdf.lower():
                self.files, self.matched = f, True
                break
        if not self.files and first is not None:
            self.files = first

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:  # noqa: BLE001
            return None

    def __enter__(self):
        if self.files:
            import threading

            def loop():
                while not self._stop:
                    self.rows.append((self._read(self.files["power"]), self._read(self.files["sclk"]) if "sclk" in self.files else None))
                    time.sleep(self.period)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)

    def summary(self):
        pw = [r[0] for r in self.rows if r[0] is not None]
        ck = [r[1] for r in self.rows if r[1] is not None]
        if not pw:
            return None
        cap = self._read(self.files["cap"]) if "cap" in self.files else None
        out = {"avg_w": round(sum(pw) / len(pw) / 1e6, 1), "max_w": round(max(pw) / 1e6, 1), "cap_w": (round(cap / 1e6, 1) if cap else None),
               "samples": len(pw), "source": os.path.dirname(self.files["power"]), "device_matched": bool(self.matched)}
        if ck:
            out["sclk_mhz_avg"] = round(sum(ck) / len(ck) / 1e6, 1)
            out["sclk_mhz_min"] = round(min(ck) / 1e6, 1)
        return out


def _pci_bdf(dev):
    """PCI address of the torch device ("0000:bb:dd.0"), or None when the properties do not carry it."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        return f"{int(getattr(pr, 'pci_domain_id', 0)):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
    except Exception:  # noqa: BLE001
        return None


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a)))
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:     # fail loudly: the line's n_gpus must be the number of ranks that actually ran
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.launch_check:
        if world > 1:
            dist.init_process_group(a.backend if a.backend != "nccl" or torch.cuda.is_available() else "gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            if seen != world:
                raise SystemExit(f"bench.py: {seen} of {world} ranks joined")
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_joined": seen, "backend": a.backend if world > 1 else None}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.share_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPU(s) "
                         "(--share-device runs every rank on cuda:0 for a functional check)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # 'nccl' is RCCL on ROCm
        else:
            dist.init_process_group(a.backend)
        joined = torch.ones(1, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(joined)          # the first collective: every rank is really there (RCCL communicator built)
        if int(joined.item()) != world:
            raise SystemExit(f"bench.py: {int(joined.item())} of {world} ranks joined the {a.backend} group")

    from idvs.morec_amd import engine as _engine
    from idvs.morec_amd import ops
    from idvs.morec_amd.model import BertShape, HipBertModel, Model
    if a.padded:
        _engine.UNPAD_DEFAULT = False
    from idvs.morec_amd.train_step import TrainStep

    id_tower = a.tower == "id"      # BASELINE.json configs[0]: IDRec SASRec (embedding table, no modality encoder), T/train_id.py
    vision = a.tower not in ("text", "id")
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(local_rank)}; building synthetic data")
    rng = np.random.default_rng(12345)
    if vision:   # V/train_swin_tiny.py:22-41, V/parameters.py:34-39: B=64/GPU, S=10, D=2048, 224 x 224 images
        from idvs.morec_amd.model.swin import HipSwinForImageClassification
        from idvs.morec_amd.swin_engine import SwinShape
        S, T, D = 10, 0, (2048 if a.tower != "swin_micro" else 64)
        vshape = SwinShape.named(a.tower)
        a.item_num = min(a.item_num, 4096 if a.vision_input == "resident" else 1024)   # the image catalogue lives in HBM (fp32 NCHW, what
        #                                                                                 V/run.py:201-204 uploads) or, decoded uint8, on the host
        args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                     CV_model_load=a.tower, compute_dtype=a.dtype)
    else:
        S, T, D = 20, 30, 512
        shape = BertShape.named(a.bert)
        args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                     num_words_title=T, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
                                     bert_model_load="bert_" + a.bert, word_embedding_dim=shape.hidden_size,
                                     compute_dtype=a.dtype)
        content = synth_catalog(a.item_num, T, rng)
    n_batches = a.steps + a.warmup
    ids_all = synth_batches(n_batches, a.batch, S, a.item_num, np.random.default_rng(12345 + 1000 * rank))
    counts = np.bincount(ids_all.reshape(-1), minlength=a.item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    torch.manual_seed(12345)
    tower = None if id_tower else (HipSwinForImageClassification(vshape, D) if vision else HipBertModel(shape))
    model = Model(args, a.item_num, not id_tower, tower, pop).to(dev)
    model.train()
    log("model on device; building TrainStep arenas")
    # defer_update: with a step block (fp16) the AdamW launches of step t run under the forward pass of step t + 1 (TrainStep docstring);
    # the timed region ends with a device synchronisation, so every update it issued is inside it
    ts = TrainStep(model, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=not a.no_pool,
                   dedup_items=a.dedup, defer_update=os.environ.get("MOREC_DEFER_UPDATE", "1") != "0",
                   graph=(world == 1 and a.graph and not a.dedup))
    PAD_TO = 512 if ts.graph else 0        # spare-row bucket of the unpadded token layout (one captured graph per bucket)
    use_graph = {"v": bool(ts.graph)}

    # host batches in pinned memory: what the reference's DataLoader hands to T/run.py:232-234
    host = []
    for i in range(n_batches):
        ids = torch.from_numpy(ids_all[i]).pin_memory()
        items = None if vision else (torch.from_numpy(ids_all[i].reshape(-1).copy()).pin_memory() if id_tower
                                     else torch.from_numpy(content[ids_all[i].reshape(-1)]).pin_memory())
        lm = torch.ones(a.batch, S).pin_memory()
        # the collate's share of the unpadded token layout: row offsets / packed-row indices from the attention masks (+ the rows in
        # token-id order for the word-embedding gradient), on the host,
        # uploaded with the batch (no device-side bookkeeping, no host synchronisation inside the step)
        pack = None if (vision or id_tower or a.padded) else _engine.token_packing_host(content[ids_all[i].reshape(-1), T:], content[ids_all[i].reshape(-1), :T],
                                                                                        pad_to=PAD_TO)
        host.append((ids, items, lm, pack))
    # Warm-up batch 0 = a copy of the TIMED batch with the most real tokens (the timed set itself is untouched): activation buffers
    # are sized by the batch's token count and the caching allocator cannot reuse a smaller block for a larger request, so the first
    # batch that is larger than everything before it costs fresh hipMallocs (and, near the reservation limit, a cache flush: one
    # 300-ms step in a short timed region).  With the largest one seen during warm-up every later request is a cache hit.
    if a.warmup >= 1 and host and host[0][3] is not None and os.environ.get("MOREC_BENCH_PRIME", "1") != "0":
        j = max(range(a.warmup, n_batches), key=lambda i: int(host[i][3][1].numel()))
        host[0] = host[j]
    u8_feed, u8_stats = None, None
    if vision and a.vision_input == "u8":
        # decoded images as the LMDB of dataset/HM/build_lmdb_hm.py holds them (uint8 HWC, native size), on the HOST; a batch is packed
        # back to back + page-locked by the collate thread (run.BatchPrefetcher = the reference's DataLoader workers, V/run.py:93-94),
        # crosses PCIe as uint8 and is resampled + normalised on the GPU
        from idvs.morec_amd.data_utils.images import pack_images
        from idvs.morec_amd.run import BatchPrefetcher
        NS = a.native_size
        host_imgs = np.random.default_rng(4321).integers(0, 256, (a.item_num + 1, NS, NS, 3), dtype=np.uint8)
        u8_stats = {"pack_s": 0.0, "batches": 0, "bytes": 0}

        def make_u8(i):
            t_ = time.perf_counter()
            idx = ids_all[i].reshape(-1)
            flat, meta, tabs = pack_images([host_imgs[j] for j in idx], vshape.image_size)
            u8_stats["pack_s"] += time.perf_counter() - t_
            u8_stats["batches"] += 1
            u8_stats["bytes"] = int(flat.nbytes)
            return torch.from_numpy(flat), torch.from_numpy(meta), torch.from_numpy(tabs)
    elif vision:
        gen = torch.Generator(device=dev).manual_seed(4321)
        catalog = torch.randn((a.item_num + 1, 3, vshape.image_size, vshape.image_size), device=dev, generator=gen)
        catalog[0].zero_()

    # --- per-launch instrumentation of the dominant kernel (the NT GEMM) with HIP events on the launch stream
    gemm_log = []
    real_gemm = ops.gemm_nt
    timing_on = {"v": False}

    def timed_gemm(x, w, **kw):
        if not timing_on["v"]:
            return real_gemm(x, w, **kw)
        M = kw.get("M") or x.shape[0]
        K = kw.get("K") or x.shape[1]
        N = kw.get("N") or w.shape[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_gemm(x, w, **kw)
        e1.record()
        # algorithmic HBM bytes of the launch: both operands once, every output once, the activation-derivative operand once
        es, eo = x.element_size(), out.element_size()
        byt = (M * K + N * K) * es + M * N * eo * (1 + (kw.get("aux_out") is not None) + (kw.get("dact_in") is not None))
        gemm_log.append((2.0 * M * N * K, e0, e1, str(x.dtype), byt))
        return out

    ops.gemm_nt = timed_gemm
    real_tn = ops.gemm_tn_

    def timed_tn(dy, x, out, **kw):
        if not timing_on["v"]:
            return real_tn(dy, x, out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_tn(dy, x, out, **kw)
        e1.record()
        gemm_log.append((2.0 * dy.shape[0] * dy.shape[1] * x.shape[1], e0, e1, "tn",
                         (dy.shape[0] * dy.shape[1] + x.shape[0] * x.shape[1]) * dy.element_size() + dy.shape[1] * x.shape[1] * 4))
        return r

    ops.gemm_tn_ = timed_tn
    real_dr = ops.mlp_dact_recompute

    def timed_dr(dy, w2t, x, w1, b1, **kw):
        # FLOPs: the ONE product autograd's backward does here (dY . W2); the recomputed pre-activation is this design's overhead, not work
        if not timing_on["v"]:
            return real_dr(dy, w2t, x, w1, b1, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_dr(dy, w2t, x, w1, b1, **kw)
        e1.record()
        M, K, N = dy.shape[0], dy.shape[1], w2t.shape[0]
        gemm_log.append((2.0 * M * N * K, e0, e1, str(dy.dtype), (2 * M * K + 2 * N * K + M * N) * dy.element_size()))
        return out

    ops.mlp_dact_recompute = timed_dr

    # the scoring kernels (fused in-batch CE forward / backward): HIP events around the two C-ABI calls
    ce_log = []
    ce_shapes = []      # (kind, Nr, Nc, D) per timed call
    real_ce_f, real_ce_b = ops.inbatch_ce_fwd, ops.inbatch_ce_bwd

    def timed_ce(real, kind):
        def f(desc, P, E, *rest):
            if not timing_on["v"]:
                return real(desc, P, E, *rest)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real(desc, P, E, *rest)
            e1.record()
            es = P.element_size()
            # algorithmic bytes: fwd reads P, E + 13 B of bookkeeping per column, writes 8 B per row;
            # bwd reads P, E, lse and writes dP, dE
            nr, nc, D = P.shape[0], E.shape[0], P.shape[1]
            byt = (nr + nc) * D * es + 13 * nc + 8 * nr if kind == "fwd" else 2 * (nr + nc) * D * es + 4 * nr
            ce_log.append((byt, e0, e1))
            ce_shapes.append((kind, nr, nc, D))
            return r
        return f

    ops.inbatch_ce_fwd, ops.inbatch_ce_bwd = timed_ce(real_ce_f, "fwd"), timed_ce(real_ce_b, "bwd")

    n_run = {"v": 0}
    # --vision-input u8: the device half of the input pipeline runs on its OWN stream, one batch ahead (data_utils.images.DeviceImageFeed):
    # H2D of the packed uint8 bytes + Pillow-exact resize + normalising patch im2col of batch k + 1 under the train step of batch k -- the
    # overlap the reference gets from its DataLoader workers (V/run.py:93-94).  MOREC_BENCH_INLINE_INPUT=1: the round-4 arrangement (all of it
    # on the step's stream, in front of the encoder) for an A/B.
    dev_feed, ahead = None, {"v": None}
    if u8_stats is not None and os.environ.get("MOREC_BENCH_INLINE_INPUT", "0") != "1":
        from idvs.morec_amd.data_utils.images import DeviceImageFeed
        dev_feed = DeviceImageFeed(dev, vshape.image_size, vshape.patch_size, model.compute_dtype)

    def host_u8(i):
        """the packed host batch of step i: from the collate thread when i is next in its order, else packed inline"""
        if u8_feed is not None and u8_feed["pos"] < len(u8_feed["order"]) and u8_feed["order"][u8_feed["pos"]] == i:
            _, hb = next(u8_feed["it"])          # built ahead by the collate thread (warm-up + headline region)
            u8_feed["pos"] += 1
            return hb
        return tuple(t_.pin_memory() for t_ in make_u8(i))      # the passes after the headline

    def run_step(i):
        n_run["v"] += 1
        ids, items, lm, pack = host[i]
        ids_d = ids.to(dev, non_blocking=True)
        slot = None
        if u8_stats is not None and dev_feed is not None:
            if ahead["v"] is not None and ahead["v"][0] == i:
                slot = ahead["v"][1]
            else:                                   # nothing queued for this batch (first step, or a pass that jumps around): queue it now
                slot = dev_feed.submit(*host_u8(i))
            ahead["v"] = None
            if u8_feed is not None and u8_feed["pos"] < len(u8_feed["order"]):      # the NEXT batch: queued before this step's launches
                j = u8_feed["order"][u8_feed["pos"]]
                ahead["v"] = (j, dev_feed.submit(*host_u8(j)))
            items_d = dev_feed.take(slot)
        elif u8_stats is not None:     # host uint8 batch -> H2D -> Pillow-exact resize on the GPU -> uint8 [n, R, R, 3], on the step's stream
            flat, meta, tabs = host_u8(i)
            items_d = ops.image_resize_u8_packed(flat, meta, tabs, vshape.image_size, dev)
        else:
            items_d = catalog[ids_d.view(-1)] if vision else items.to(dev, non_blocking=True)
        lm_d = lm.to(dev, non_blocking=True)
        pack_d = None if pack is None else tuple(t.to(dev, non_blocking=True) for t in pack)
        if use_graph["v"]:
            loss = ts.step_graphed(ids_d.view(-1), items_d, lm_d, token_packing=pack_d)
        else:
            loss = ts.step(ids_d.view(-1), items_d, lm_d, token_packing=pack_d)
        if slot is not None:
            dev_feed.release(slot)
        return loss

    def start_feed(order):
        """(u8 input) a collate thread that builds the batches `order` names, two ahead of the device"""
        nonlocal u8_feed
        if u8_stats is None:
            return
        if u8_feed is not None:
            u8_feed["feeder"].close()
        feeder = BatchPrefetcher(lambda i: make_u8(i), list(order), depth=2)
        u8_feed = {"feeder": feeder, "it": iter(feeder), "order": list(order), "pos": 0}

    # graph mode: every input shape of the timed batches is seen twice before the clock starts (first sight runs eagerly, the second
    # is the capture) -- the per-shape analogue of priming the allocator with the largest batch
    if use_graph["v"] and u8_stats is None:
        shapes = {}
        for i in range(a.warmup, n_batches):
            shapes.setdefault(None if host[i][3] is None else int(host[i][3][1].numel()), i)
        log(f"graph mode: {len(shapes)} input shape(s) among the timed batches; capturing")
        for i in shapes.values():
            run_step(i)
            run_step(i)
    start_feed(list(range(a.warmup)) + list(range(a.warmup, n_batches)))
    log("warm-up")
    for i in range(a.warmup):
        loss = run_step(i)
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss {float(loss.item()):.4f}")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the headline region: exactly K steps, NO instrumentation inside (no events, no host reads)
    t0 = time.perf_counter()
    for i in range(a.warmup, n_batches):
        loss = run_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_v = float(loss.item())
    log(f"timed region done: {dt / a.steps * 1e3:.2f} ms/step")
    # ---- N > 1: the headline is measured; everything from here on (traced step, reduced sweep, instrumented pass) is explanation.  No
    # multi-GPU run of this file has ever happened on hardware, and a collective that hangs in that part would take the measured line
    # with it: a watchdog prints the headline alone and ends every rank if the rest does not finish in time (MOREC_BENCH_WATCHDOG_S,
    # default 240 s; 0 = off).
    watchdog = None
    if world > 1:
        wd_s = float(os.environ.get("MOREC_BENCH_WATCHDOG_S", "240"))
        if wd_s > 0:
            import threading
            wd_line = {"metric": "user-sequences/sec end-to-end train step, SASRec+BERT-base", "value": round(world * a.batch * a.steps / dt, 2),
                       "unit": "user-seq/s", "n_gpus": world, "world_size_observed": dist.get_world_size(), "steps": a.steps, "warmup": a.warmup,
                       "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                       "dtype": a.dtype, "data": "synthetic MIND-shaped (80k items, Zipf(1.0) popularity, 30-token titles, history 23), random-init weights",
                       "config": {"workload": f"SASRec(2 blocks, 2 heads, D=512) + BERT-{a.bert} text encoder, in-batch debiased CE, B={a.batch}/GPU, S=20, T=30",
                                  "global_batch": world * a.batch, "seq_len": S + 3,
                                  "parallelism": f"dp{world}" + ("" if a.no_pool else "+pooled-negatives")},
                       "final_loss": round(loss_v, 4), "roofline": None, "cpu_baseline": None,
                       "post_headline": f"watchdog: the traced step / sweep / instrumented pass behind the timed region did not finish within {wd_s:.0f} s; "
                                        "the headline region itself completed on every rank (barrier + max over ranks)"}

            def _wd_fire():
                if rank == 0:
                    print(json.dumps(wd_line), flush=True)
                os._exit(0)
            watchdog = threading.Timer(wd_s, _wd_fire)
            watchdog.daemon = True
            watchdog.start()
    u8_line = None
    if u8_stats is not None:
        u8_line = {"native_size": a.native_size, "uint8_bytes_per_step": u8_stats["bytes"],
                   "host_pack_ms_per_batch": round(u8_stats["pack_s"] / max(1, u8_stats["batches"]) * 1e3, 2),
                   "collate": "one thread, two batches ahead (run.BatchPrefetcher), page-locked",
                   "in_timed_region": "H2D of the packed uint8 batch + morec_image_resize_u8 + morec_swin_patchify_u8 (ToTensor + Normalize fused) + the train step",
                   "device_side": ("own HIP stream, one batch ahead of the step (data_utils.images.DeviceImageFeed: double-buffered, one event per batch)"
                                   if dev_feed is not None else "on the step's stream, in front of the encoder (MOREC_BENCH_INLINE_INPUT=1)")}
        if u8_feed is not None:
            u8_feed["feeder"].close()
            u8_feed = None
    # ---- N > 1: one traced step -- when each gradient bucket's all-reduce is issued (stream time from the start of the step) and how
    # long the closing join waits, so that a scaling run explains itself
    def traced_step():
        """One step with stream-time events: when each gradient bucket's collective is issued (and, on the library's own communicator,
        when it completes), when the closing join starts and ends (= the exposed wait), and the scoring calls at the pooled size."""
        try:
            ts.trace = []
            del ce_log[:], ce_shapes[:]
            timing_on["v"] = True
            e_start = torch.cuda.Event(enable_timing=True)
            e_start.record()
            run_step(a.warmup)
            e_end = torch.cuda.Event(enable_timing=True)
            e_end.record()
            torch.cuda.synchronize()
            timing_on["v"] = False
            ev_list, ts.trace = ts.trace, None
            tr = {"step_ms": round(e_start.elapsed_time(e_end), 3), "buckets": []}
            for rec in ev_list:
                if rec[0] == "issue":
                    _, gi, lo, hi, ev = rec
                    tr["buckets"].append({"group": gi, "MB": round((hi - lo) * 4 / 1e6, 1), "issued_at_ms": round(e_start.elapsed_time(ev), 3)})
                elif rec[0] == "done":      # (own RCCL communicator on a side stream: the collective's completion in stream time)
                    _, gi, lo, hi, ev = rec
                    for bk in tr["buckets"]:
                        if bk["group"] == gi and bk["MB"] == round((hi - lo) * 4 / 1e6, 1) and "completed_at_ms" not in bk:
                            bk["completed_at_ms"] = round(e_start.elapsed_time(ev), 3)
                            break
                else:
                    tr[rec[0] + "_ms"] = round(e_start.elapsed_time(rec[1]), 3)
            if "join_begin_ms" in tr and "join_end_ms" in tr:
                tr["exposed_join_ms"] = round(tr["join_end_ms"] - tr["join_begin_ms"], 3)
            if ce_log:
                tr["pooled_scoring"] = {"Nr": ce_shapes[0][1], "Nc": ce_shapes[0][2], "D": ce_shapes[0][3],
                                        "fwd_us": round(ce_log[0][1].elapsed_time(ce_log[0][2]) * 1e3, 1),
                                        "bwd_us": round(ce_log[-1][1].elapsed_time(ce_log[-1][2]) * 1e3, 1) if len(ce_log) > 1 else None}
            return tr
        except Exception as e:  # noqa: BLE001
            timing_on["v"] = False
            ts.trace = None
            return {"error": f"{type(e).__name__}: {e}"}

    # ---- N > 1: one traced step, so that a scaling run explains itself
    keep_ce = (list(ce_log), list(ce_shapes))
    reduce_trace = traced_step() if world > 1 else None
    # ---- N > 1, --sweep: the knobs of the data-parallel step measured in ONE launch, so that the first multi-GPU run settles them:
    # {torch.distributed | the library's own RCCL communicators} x {CUs kept out of the GEMM grid during the backward: 0, 8, 16} x
    # {bucketed reduction overlapped with the backward | one sweep after it}.  Every configuration: 2 untimed + K timed steps between
    # barriers (max over ranks), then one traced step.  Schema: INTEGRATION.md "bench.py --sweep".
    # Without --sweep (the driver's scaling runs) a REDUCED sweep still runs at N > 1: torch.distributed only -- no second communicator is
    # created, nothing that could hang a run that has never been on > 1 GPU -- x {overlap, reserve 16 | overlap, reserve 0 | after the
    # backward}, so that the first real multi-GPU line already says what the reserved CUs and the bucket overlap are worth (--no-sweep: off).
    sweep = None
    if world > 1 and (a.sweep or not a.no_sweep):
        sweep = []
        full = bool(a.sweep)
        base = (ts.comm, ts.comm_grad, ts._grad_stream, ts.reserve_cus, ts.overlap_reduce)
        own = None
        if not full:
            pass
        elif a.backend == "nccl" and not a.share_device:
            try:
                from idvs.morec_amd.comm import MorecComm
                own = (MorecComm(), MorecComm(), torch.cuda.Stream(device=dev))
                probe = torch.ones(1, device=dev)
                own[0].all_reduce_sum_(probe)
                if int(probe.item()) != world:
                    raise SystemExit(f"bench.py --sweep: the library's RCCL communicator sees {int(probe.item())} of {world} ranks")
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                own = None
                sweep.append({"comm": "rccl", "skipped": f"{type(e).__name__}: {e}"})
        else:
            sweep.append({"comm": "rccl", "skipped": "needs backend nccl with one GPU per rank (RCCL refuses two ranks on one device)"})
        K = max(2, min(a.steps, 6 if full else 4))
        for comm_name in ("torch.distributed", "rccl"):
            if comm_name == "rccl" and own is None:
                continue
            for overlap in (True, False):
                for reserve in (((0, 8, 16) if full else (0, 16)) if (overlap and a.backend == "nccl") else (0,)):
                    ts.comm, ts.comm_grad, ts._grad_stream = (own if comm_name == "rccl" else (None, None, None))
                    ts.overlap_reduce, ts.reserve_cus = overlap, reserve
                    rec = {"comm": comm_name, "overlap_reduce": overlap, "reserve_cus": reserve}
                    try:
                        for i in range(2):
                            run_step(i)
                        dist.barrier()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for i in range(K):
                            run_step(a.warmup + i % a.steps)
                        torch.cuda.synchronize()
                        dist.barrier()
                        tt = torch.tensor([time.perf_counter() - t1], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        rec.update({"steps": K, "ms_per_step": round(float(tt.item()) / K * 1e3, 3),
                                    "user_seq_per_s": round(world * a.batch * K / float(tt.item()), 2), "trace": traced_step()})
                    except Exception as e:  # noqa: BLE001
                        rec["error"] = f"{type(e).__name__}: {e}"
                    sweep.append(rec)
        ts.comm, ts.comm_grad, ts._grad_stream, ts.reserve_cus, ts.overlap_reduce = base
        ts._reserve(False)
    ce_log[:], ce_shapes[:] = keep_ce
    # ---- instrumented pass over the same batches (after the headline; never part of `value`): HIP events around every GEMM /
    # scoring call on the launch stream -> roofline.achieved
    # The weight-gradient stream is switched OFF for this pass: with it on, dW launches overlap the dX chain and the event-timed
    # durations of both stretch over each other (they would add up to more than the step).  The pass therefore measures each GEMM
    # launch with the chip to itself, i.e. the kernels' own rate; the headline region above runs with the overlap.
    n_inst = min(a.steps, 8)
    graph_was, use_graph["v"] = use_graph["v"], False      # the per-launch events need the launches: every pass from here on is eager
    wgrad_was = _engine.WgradStream.enabled
    _engine.WgradStream.enabled = False
    run_step(a.warmup)
    torch.cuda.synchronize()
    timing_on["v"] = True
    for i in range(a.warmup, a.warmup + n_inst):
        run_step(i)
    torch.cuda.synchronize()
    timing_on["v"] = False
    _engine.WgradStream.enabled = wgrad_was
    # ---- sustained rate: the same K batches cycled for >= 3 s (the shader clock of a power-limited part settles over seconds;
    # the K-step headline region is < 1 s)
    sustained = None
    if not a.no_secondary and world == 1:
        use_graph["v"] = graph_was
        n_sus, t1 = 0, time.perf_counter()
        with PowerSampler(pci_bdf=_pci_bdf(dev)) as psamp:
            while True:
                for i in range(a.warmup, n_batches):
                    run_step(i)
                n_sus += a.steps
                torch.cuda.synchronize()
                if time.perf_counter() - t1 >= 3.0:
                    break
            dts = time.perf_counter() - t1
        sustained = {"ms_per_step": round(dts / n_sus * 1e3, 3), "user_seq_per_s": round(a.batch * n_sus / dts, 2), "steps": n_sus,
                     "seconds": round(dts, 2), "note": "the K batches of the headline region cycled back to back for >= 3 s",
                     "power": psamp.summary()}
        log(f"sustained: {sustained['ms_per_step']} ms/step over {n_sus} steps")
    use_graph["v"] = False
    # secondary measurement (never `value`): the same steps with distinct-item dedup on, and the duplicate rate of the batches
    def timed_again():
        for i in range(min(2, a.warmup)):
            run_step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.warmup, n_batches):
            run_step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t1

    padded_info = None
    if not vision and not id_tower and not a.padded and not a.no_secondary and _engine.UNPAD_DEFAULT:
        _engine.UNPAD_DEFAULT = False
        dt3 = timed_again()
        _engine.UNPAD_DEFAULT = True
        real_catalog = float((content[1:, T:] != 0).mean())
        real_batches = float(np.mean([(content[ids_all[i].reshape(-1), T:] != 0).mean() for i in range(a.warmup, n_batches)]))
        padded_info = {"ms_per_step": round(dt3 / a.steps * 1e3, 3), "user_seq_per_s_this_rank_clock": round(world * a.batch * a.steps / dt3, 2),
                       "real_token_fraction": round(real_batches, 4), "real_token_fraction_catalog_mean": round(real_catalog, 4),
                       "note": "--padded: encoder layers over all 30 positions of every title, as the reference computes them; the default "
                               "runs them on the real tokens only ([PAD] keys have probability exactly 0 and only hidden[:, 0] is consumed: same item vectors). "
                               "real_token_fraction = real tokens / (Nc T) over the timed BATCHES (Zipf sampling favours some titles); "
                               "..._catalog_mean = the same over the catalog"}
    dedup_info = None
    if not a.dedup and not a.no_secondary and not id_tower:
        ts.dedup_items = True
        for i in range(min(2, a.warmup)):
            run_step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(a.warmup, n_batches):
            run_step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt2 = time.perf_counter() - t1
        ts.dedup_items = False
        uniq = float(np.mean([len(np.unique(ids_all[i])) for i in range(n_batches)])) / float(ids_all[0].size)
        dedup_info = {"ms_per_step": round(dt2 / a.steps * 1e3, 3), "user_seq_per_s_this_rank_clock": round(world * a.batch * a.steps / dt2, 2),
                      "distinct_item_fraction": round(uniq, 4),
                      "note": "opt-in (--dedup): each distinct item of the batch encoded once; exact with dropout off, shares dropout masks between duplicates otherwise"}

    # secondary line (never `value`): the same step in the PARITY mode (exact-fp32 MFMA, v_mfma_f32_16x16x4_f32 -- the mode the
    # reference goldens pin at 1e-4 on the loss), so that the price of reference-level numerics is a measured number
    fp32_info = fp32x3_info = None
    main_gemm_log = list(gemm_log)
    main_ce_log, main_ce_shapes = list(ce_log), list(ce_shapes)
    if not vision and not id_tower and a.dtype16 in ("bf16", "fp16") and not a.no_secondary and world == 1:
        def fp32_mode_line(mode):
            nonlocal ts
            info = None
            saved = (ts, )
            try:
                args32 = types.SimpleNamespace(**dict(vars(args), compute_dtype=mode))
                torch.manual_seed(12345)
                model32 = Model(args32, a.item_num, True, HipBertModel(shape), pop).to(dev)
                model32.train()
                ts32 = TrainStep(model32, lr=1e-4, fine_tune_lr=5e-5, l2_weight=0.01, fine_tune_l2_weight=0.01, pool_negatives=False)
                ts = ts32                      # run_step closes over `ts`
                n32 = max(2, min(4, a.steps))
                for _ in range(3 if mode == "fp32x3" else 1):      # fp32x3 holds ~90 GB of splits per step: let the caching allocator reach its size first
                    run_step(0)
                torch.cuda.synchronize()
                del gemm_log[:]
                timing_on["v"] = True
                t1 = time.perf_counter()
                for i in range(n32):
                    run_step(a.warmup + i % a.steps)
                torch.cuda.synchronize()
                dt32 = time.perf_counter() - t1
                timing_on["v"] = False
                fl32 = sum(g_[0] for g_ in gemm_log)
                ms32 = sum(g_[1].elapsed_time(g_[2]) for g_ in gemm_log)
                info = {"ms_per_step": round(dt32 / n32 * 1e3, 2), "user_seq_per_s": round(a.batch * n32 / dt32, 2), "steps": n32}
                if mode == "fp32":
                    tf32 = fl32 / (ms32 * 1e-3) / 1e12 if ms32 > 0 else 0.0
                    info.update({"gemm_tflops": round(tf32, 1), "mfma_f32_peak": MFMA_PEAK_TFLOPS["f32"],
                                 "frac_of_f32_mfma_peak": round(tf32 / MFMA_PEAK_TFLOPS["f32"], 4),
                                 "note": "compute_dtype=fp32: every GEMM on exact-fp32 MFMA; the mode whose loss matches the reference goldens "
                                         "to < 1e-4 (tests/test_model_gpu.py g6). tests/test_bench_mode_parity_gpu.py bounds the bf16 mode "
                                         "against it at this configuration: step-0 loss 3e-2 (measured 1.3e-3 ... 1.5e-2: rounding-pattern dependent), gradient norms 5e-2 (measured 0.6e-2 ... 2.3e-2), 20-step loss curve 2 % (measured 0.9 %)"}) 
                else:
                    info["note"] = ("compute_dtype=fp32x3: fp32 tensors, every GEMM as ONE bf16 MFMA product over hi / lo splits of both operands "
                                    "(morec_split_bf16x3, K' = 3 K, fp32 accumulation; the lo.lo term, 2^-16 relative, dropped): against the exact-fp32 "
                                    "mode step-0 loss 5e-7 relative, gradient norms 4e-5, loss curve 1e-5 (tests/test_fp32x3_gpu.py asserts 2e-4 / 1e-3 / "
                                    "1e-3) -- inside north_star's 1e-3, which the bf16 mode is not")
                del ts32, model32
            except Exception as e:  # noqa: BLE001 -- a secondary line must never cost the headline
                info = {"error": f"{type(e).__name__}: {e}"}
                timing_on["v"] = False
            (ts, ) = saved
            torch.cuda.empty_cache()
            return info
        fp32_info = fp32_mode_line("fp32")
        fp32x3_info = fp32_mode_line("fp32x3")
        gemm_log[:] = main_gemm_log
        ce_log[:] = main_ce_log
        ce_shapes[:] = main_ce_shapes

    # roofline of the dominant kernel: algorithmic FLOPs of every GEMM launch / its measured duration
    fl = sum(g_[0] for g_ in gemm_log)
    ms = sum(g_[1].elapsed_time(g_[2]) for g_ in gemm_log)
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # fp32x3: three bf16 MFMA passes per fp32 product -> a third of the bf16 peak in fp32-equivalent FLOPs
    peak = {"bf16": MFMA_PEAK_TFLOPS["bf16"], "fp16": MFMA_PEAK_TFLOPS["fp16"], "fp32": MFMA_PEAK_TFLOPS["f32"],
            "fp32x3": round(MFMA_PEAK_TFLOPS["bf16"] / 3.0, 1)}[a.dtype16]
    launches_per_step = len(gemm_log) / max(1, n_inst)
    flops_per_step = fl / max(1, n_inst)
    alg_bytes_per_launch = sum(g_[4] for g_ in gemm_log) / max(1, len(gemm_log))
    # HBM bytes per launch from the rocprofv3 PMC passes (scripts/capture_profiles.sh -> profiles/<round>_gemm_pmc.json).  The file is
    # used only when it describes THIS workload: same number of GEMM launches per step, same token layout, FLOPs per step within 3 %
    # (the profiled run draws fewer batches of the same synthetic set)
    traffic, traffic_note = None, "no PMC file under profiles/ matches this run"
    pmc_files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gemm_pmc.json")), reverse=True) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    layout_now = None if (vision or id_tower) else ("padded (all T positions)" if (a.padded or not _engine.UNPAD_DEFAULT) else "unpadded (real tokens only; exact)")
    for fn in pmc_files:
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                pj = json.load(fh)
            sig = pj.get("signature") or {}
            # (the profile counts GEMM KERNELS, this file counts ops.gemm_* calls: the two GEMMs inside morec_inbatch_ce_bwd are the difference)
            if (sig.get("gemm_launches_per_step") is not None and abs(sig["gemm_launches_per_step"] - launches_per_step) < 3
                    and sig.get("token_layout") == layout_now and sig.get("gemm_flops_per_step")
                    and abs(sig["gemm_flops_per_step"] / flops_per_step - 1.0) < 0.03):
                traffic, traffic_note = pj.get("hbm_bytes_per_launch_avg"), f"profiles/{fn}: " + pj.get("correction", "")
                break
        except Exception:  # noqa: BLE001
            continue
    roof = {"bound": "mfma", "kernel": "gemm8p_kernel + gemm_tn8p_kernel (256 x 256 eight-phase MFMA 32x32x16 tiles) + the small-problem gemm_nt / gemm_tn "
                                       "kernels: every GEMM launch of the step",
            "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": round(alg_bytes_per_launch),
            "traffic_over_algorithmic": (round(traffic / alg_bytes_per_launch, 3) if traffic else None),
            "launches_per_step": int(round(launches_per_step)), "gemm_ms_per_step": round(ms / max(1, n_inst), 3),
            "gemm_flops_per_step": round(flops_per_step),
            "measured": f"HIP events around every GEMM launch in an instrumented pass of {n_inst} steps AFTER the headline region (which carries no events); "
                        "single stream in that pass (the headline region overlaps the weight-gradient GEMMs with the dX chain on a second stream: "
                        + ("on" if wgrad_was else "off") + ")"}
    if sustained is not None and sustained.get("power") and sustained["power"].get("sclk_mhz_avg") and sustained["power"].get("device_matched"):
        ck = sustained["power"]["sclk_mhz_avg"]
        roof["at_sustained_clock"] = {"sclk_mhz": ck, "peak": round(peak * ck / 2400.0, 1), "frac": round(tf / (peak * ck / 2400.0), 4),
                                      "note": "the 2.5 PFLOP/s peak is 256 CUs x 4096 FLOP/clk at 2.4 GHz; under this step's load the part holds its power cap at the "
                                              "shader clock sampled during `sustained` (hwmon freq1_input), which scales what the matrix cores can issue"}
    ce_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in ce_log)
    ce_gbs = sum(b for b, _, _ in ce_log) / (ce_ms * 1e-3) / 1e9 if ce_ms > 0 else 0.0
    # Scoring is MFMA-bound, not HBM-bound: 2 Nr Nc D FLOP per product (1 forward, 3 backward: recompute, dP, dE) over
    # (Nr + Nc) D bytes of operands = 1300 FLOP/B at the one-GPU size, 2260 at the 8-rank pooled size, against a machine balance
    # of 2.5e15 / 8e12 = 312 (profiles/r02_scoring_pooled.txt; the logits are never stored).  The algorithmic byte rate is kept for
    # the record.
    ce_fl = sum((2.0 if kind == "fwd" else 6.0) * nr * nc * dd for kind, nr, nc, dd in ce_shapes)
    ce_tf = ce_fl / (ce_ms * 1e-3) / 1e12 if ce_ms > 0 else 0.0
    roof["scoring"] = {"bound": "mfma", "kernel": "ce8p_kernel<fwd> + ce_combine / ce8p_kernel<bwd> + gemm8p (dE = dl^T P) + gemm_tn8p (dP = dl E) (fused in-batch debiased CE on 256 x 256 "
                                                   "eight-phase tiles; logits never stored)",
                       "achieved": round(ce_tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ce_tf / peak, 4),
                       "algorithmic_GBps": round(ce_gbs, 1), "ms_per_step": round(ce_ms / max(1, n_inst), 4),
                       "note": "launch-latency class at the one-GPU size (2560 x 2688 logits: 110 tiles for 256 CUs); the size the multi-GPU step runs is "
                               "`pooled_8_ranks`"}

    # the scoring kernels at the 8-rank POOLED size (this rank's B S rows against eight ranks' worth of item vectors, emulated on one
    # GPU): the size north_star's multi-GPU step runs them at; never part of `value`
    if not a.no_secondary and not vision and world == 1 and a.dtype16 in ("bf16", "fp16"):
        try:
            roof["scoring"]["pooled_8_ranks"] = scoring_pooled(ops, a.batch, S, D, dev, peak, dt=torch.float16 if a.dtype16 == "fp16" else torch.bfloat16)
        except Exception as e:  # noqa: BLE001
            roof["scoring"]["pooled_8_ranks"] = {"error": f"{type(e).__name__}: {e}"}

    # Is the GEMM rate set by the schedule or by the part's power limit?  The same launch (FFN-up shape of the step, plain epilogue) on N(0, 0.5)
    # operands and on zeros: identical instruction stream and cycle count, different switching activity (never part of `value`)
    if not a.no_secondary and not vision and not id_tower and world == 1 and a.dtype16 in ("bf16", "fp16"):
        try:
            roof["operand_activity_probe"] = operand_activity_probe(ops, dev, torch.float16 if a.dtype16 == "fp16" else torch.bfloat16)
        except Exception as e:  # noqa: BLE001
            roof["operand_activity_probe"] = {"error": f"{type(e).__name__}: {e}"}

    eval_info = None
    if not a.no_secondary and not vision and not id_tower and world == 1 and a.dtype16 in ("bf16", "fp16"):
        keep = (list(gemm_log), list(ce_log), list(ce_shapes))
        eval_info = eval_lines(model, ops, args, content, a.item_num, S, D, dev, gemm_log, timing_on, peak)
        gemm_log[:], ce_log[:], ce_shapes[:] = keep
    scaler_state = None
    if ts.sp is not None:
        h_ = ts.sp.host()
        scaler_state = {"scale": float(h_.loss_scale), "applied_steps": int(h_.step), "skipped_steps": int(h_.skipped)}

    out = {"metric": "user-sequences/sec end-to-end train step, SASRec+BERT-base", "value": round(world * a.batch * a.steps / dt, 2),
           "unit": "user-seq/s", "n_gpus": world, "world_size_observed": (dist.get_world_size() if world > 1 else 1), "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.dtype, "data": "synthetic MIND-shaped (80k items, Zipf(1.0) popularity, 30-token titles, history 23), random-init weights",
           "config": {"weight_gradient_stream": bool(_engine.WgradStream.enabled),
                      "workload": f"SASRec(2 blocks, 2 heads, D=512) + BERT-{a.bert} text encoder, in-batch debiased CE, "
                                  f"B={a.batch}/GPU, S=20, T=30", "global_batch": world * a.batch, "seq_len": S + 3,
                      "parallelism": f"dp{world}" + ("" if a.no_pool or world == 1 else "+pooled-negatives"),
                      "dropout": "on (p = 0.1 hidden + attention, SASRec and BERT; counter-based masks fused in the kernels)"},
           "final_loss": round(loss_v, 4), "roofline": roof, "steps_executed": n_run["v"]}
    if sustained is not None:
        out["sustained"] = sustained
    if eval_info is not None:
        out["eval"] = eval_info
    if scaler_state is not None:
        out["loss_scaler_state"] = scaler_state
    if reduce_trace is not None:
        out["gradient_reduce_trace"] = reduce_trace
    if sweep is not None:
        out["sweep"] = sweep
    if world > 1:
        out["config"]["gemm8p_reserve_cus"] = int(ts.reserve_cus)      # CUs kept out of the GEMM grid during the backward pass (0 without an RCCL ring)
    if id_tower:
        out["metric"] = "user-sequences/sec end-to-end train step, IDRec SASRec (embedding table)"
        out["config"]["workload"] = (f"SASRec(2 blocks, 2 heads, D=512) + ID embedding table ({a.item_num} items, dense AdamW over the table), "
                                     f"in-batch debiased CE, B={a.batch}/GPU, S=20")
    out["config"]["launch"] = (f"hipGraph replay: one captured graph per input shape ({len([v for v in ts._graphs.values() if v != 'seen'])} captured"
                               + (f", unpadded token rows padded to multiples of {PAD_TO}" if PAD_TO and not vision and not id_tower and not a.padded else "") + ")"
                               if graph_was else "host launches (eager)")
    if not vision and not id_tower:
        out["config"]["token_layout"] = "padded (all T positions)" if (a.padded or not _engine.UNPAD_DEFAULT) else "unpadded (real tokens only; exact)"
    if padded_info is not None:
        out["padded_token_layout"] = padded_info
    if dedup_info is not None:
        out["with_item_dedup"] = dedup_info
    # secondary lines (never `value`): the other BASELINE.json configurations, each measured by a child run of this file so that the
    # driver's record carries a time for every config: configs[3] Swin-T (V/train_swin_tiny.py:22-41: B = 64/GPU, 704 images per
    # step), configs[4] Swin-B (B = 32/GPU, 352 images per step), configs[0] IDRec (T/train_id.py:22-26), configs[1] BERT-tiny
    if not vision and not id_tower and a.dtype16 in ("bf16", "fp16") and not a.no_secondary and world == 1 and a.bert == "base":
        import subprocess

        def child(extra, timeout=240):
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-secondary"],
                               capture_output=True, text=True, timeout=timeout)
            return json.loads(r.stdout.strip().splitlines()[-1])

        other16 = "bf16" if a.dtype16 == "fp16" else "fp16"
        res32 = a.dtype16 + "_res32" if a.dtype == a.dtype16 else a.dtype16
        for key, extra, per_seq in ((other16 + "_mode", ["--dtype", other16, "--batch", str(a.batch), "--steps", "20", "--warmup", "5"], 0),
                                    (res32 + "_mode", ["--dtype", res32, "--batch", str(a.batch), "--steps", "20", "--warmup", "5"], 0),
                                    ("vision_swin_tiny", ["--tower", "swin_tiny", "--batch", "64", "--steps", "6", "--warmup", "2"], 11),
                                    ("vision_swin_base", ["--tower", "swin_base", "--batch", "32", "--steps", "4", "--warmup", "2"], 11),
                                    # (BASELINE.json configs[4] names the bf16 MFMA path for Swin-B: the same child in the other 16-bit dtype)
                                    ("vision_swin_base_" + other16, ["--tower", "swin_base", "--batch", "32", "--steps", "4", "--warmup", "2", "--dtype", other16], 11),
                                    ("vision_u8_pipeline", ["--tower", "swin_tiny", "--batch", "64", "--steps", "6", "--warmup", "2", "--vision-input", "u8"], 11),
                                    # (the two latency-class configurations: 1 - 2 ms per step, so 20 timed steps are 30 ms of wall clock and one host hiccup is 50 % -- 100 steps)
                                    ("id_tower", ["--tower", "id", "--batch", "128", "--steps", "100", "--warmup", "20"], 0),
                                    ("bert_tiny", ["--bert", "tiny", "--batch", "128", "--steps", "100", "--warmup", "20"], 0)):
            try:
                vj = child(extra if "--dtype" in extra else extra + ["--dtype", a.dtype16])
                out[key] = {"ms_per_step": vj["ms_per_step"], "user_seq_per_s": vj["value"], "dtype": vj["dtype"], "config": vj["config"]["workload"],
                            "gemm_roofline_frac": vj.get("roofline", {}).get("frac"),
                            "note": "python bench.py " + " ".join(extra)}
                if per_seq:
                    out[key]["images_per_s"] = round(vj["value"] * per_seq, 1)
                if "vision_input_pipeline" in vj:
                    out[key]["input"] = vj["vision_input_pipeline"]
            except Exception as e:  # noqa: BLE001 -- a secondary line must never cost the headline
                out[key] = {"error": f"{type(e).__name__}: {e}"}
    TOL = {"fp16": "fp16 vs the exact-fp32 parity mode AT THIS CONFIGURATION (tests/test_fp16_mode_gpu.py, asserted): step-0 loss 1e-3 relative "
                   "(measured 3.1e-4; 3.4e-3 absolute on 10.90), gradient norms 1.5e-2 (measured 7.9e-3), 20-step loss curve 1e-2 relative; reference "
                   "golden g6 BERT-base loss 1e-3 relative (measured 5.3e-4); HR@10 of the modal eval golden g17 equal to the reference's",
           "bf16": "bf16 vs the exact-fp32 parity mode at this configuration (tests/test_bench_mode_parity_gpu.py, asserted): step-0 loss 3e-2 absolute "
                   "(measured 1.3e-3 ... 1.5e-2 depending on the GEMM summation order), gradient norms 5e-2 (0.6e-2 ... 2.3e-2), 20-step loss curve 2 %"}
    TOL["fp16_res32"] = ("fp16 GEMM operands / outputs with an fp32 residual stream (LayerNorm in and out fp32): the data flow of the reference's "
                         "torch.cuda.amp.autocast step (T/run.py:242); against the exact-fp32 mode at this configuration: tests/test_fp16_mode_gpu.py")
    if a.dtype in TOL and not vision and not id_tower:
        out["config"]["tolerance_vs_fp32_mode"] = TOL[a.dtype]
    TOL_V = {"fp16": "fp16 vs the exact-fp32 parity mode, Swin-T at 176 images (tests/test_bench_mode_parity_vision_gpu.py, asserted): step-0 loss 4e-3 "
                     "(measured 7e-6), gradient norms 1e-2 (1.4e-3), steps 0-4 within 1 % of the loss (measured 0.17 %); reference goldens g13 / g15 (full-size Swin-T / -B): "
                     "loss 6e-3 (measured 1.5e-3 / 1.7e-3)",
             "bf16": "bf16 vs the exact-fp32 parity mode, Swin-T at 176 images (same file): step-0 loss 3e-2 (measured 1.2e-2), gradient norms 5e-2 (1.1e-2)"}
    if vision and a.dtype in TOL_V:
        out["config"]["tolerance_vs_fp32_mode"] = TOL_V[a.dtype]
    if a.dtype16 == "fp16":
        out["config"]["loss_scaling"] = ("GradScaler protocol on the device (morec_step_params: init 65536, x0.5 + skipped step on inf / NaN, x2 after 2000 clean "
                                         "steps); the overflow check, the decision and AdamW are inside the timed step")
    if isinstance(out.get("fp16_res32_mode"), dict) and "user_seq_per_s" in out["fp16_res32_mode"]:
        # the same step in the mode that meets north_star's 1e-3 as an ABSOLUTE bound on the loss (fp32 residual stream = the data flow of the
        # reference's autocast; `value` above is the plain fp16 mode: 3e-4 relative, 3.4e-3 absolute at this configuration)
        r32 = out["fp16_res32_mode"]
        out["value_at_1e-3_abs"] = {"value": r32["user_seq_per_s"], "ms_per_step": r32["ms_per_step"], "dtype": "fp16_res32",
                                    "tolerance": "step-0 loss within 1e-3 ABSOLUTE of the exact-fp32 mode at this configuration (measured 2.9e-4), gradient norms "
                                                 "8.7e-4, 20-step curve 3.8e-4: tests/test_fp16_mode_gpu.py"}
    if fp32_info is not None:
        out["fp32_parity_mode"] = fp32_info
        out["fp32x3_mode"] = fp32x3_info
    if a.dedup:
        out["config"]["item_dedup"] = True
    if u8_line is not None:
        out["vision_input_pipeline"] = u8_line
    if vision:
        out["metric"] = f"user-sequences/sec end-to-end train step, SASRec+{a.tower}"
        out["data"] = (f"synthetic HM-shaped ({a.item_num} items, {vshape.image_size}x{vshape.image_size} fp32 images resident in HBM, "
                       "history 11), random-init weights") if u8_line is None else (
                       f"synthetic HM-shaped ({a.item_num} items, decoded uint8 {a.native_size}x{a.native_size} images on the HOST, uploaded and "
                       f"resampled to {vshape.image_size}x{vshape.image_size} on the GPU every step, history 11), random-init weights")
        out["config"] = {"workload": f"SASRec(2 blocks, 2 heads, D={D}) + {a.tower} vision encoder, in-batch debiased CE, "
                                     f"B={a.batch}/GPU, S={S}, {a.batch * (S + 1)} images/GPU/step", "global_batch": world * a.batch,
                         "seq_len": S + 1, "parallelism": out["config"]["parallelism"],
                         "dropout": "on (SASRec p = 0.1; Swin DropPath 0 -> 0.1 linear, per-image scales from the counter-based RNG)"}

    if rank == 0 and not a.no_cpu_baseline and (vision or id_tower):
        out["cpu_baseline"] = {"value": None, "unit": "user-seq/s", "cores": None, "kind": "port",
                               "sample": "not measured for the vision tower (the BASELINE.json metric is the text tower)"}
    elif rank == 0 and not a.no_cpu_baseline:
        # the CPU oracle runs in a child process under a wall-clock limit so it can never block the result line
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--bert", a.bert,
                                "--item-num", str(a.item_num), "--cpu-batch", str(a.cpu_batch)], capture_output=True,
                               text=True, timeout=a.cpu_timeout, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "user-seq/s", "cores": None, "kind": "port",
                                   "sample": f"not measured: {type(e).__name__}"}
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))      # the line is out: a closing barrier that hangs must not turn the run into a failure
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


def eval_lines(model, ops, args, content, item_num, S, D, dev, gemm_log, timing_on, peak, n_users=16384, test_bs=4096):
    """The evaluation pass of an epoch (T/data_utils/metrics.py:60-107; never part of `value`): (1) encode ALL items -- `get_item_embeddings`
    over the whole synthetic catalogue, host token rows uploaded chunk by chunk as the reference does -- items/s and the MFMA rate of its
    GEMMs; (2) `eval_model`'s ranking: user states from the SASRec encoder + `morec_eval_rank` (count-greater over exact-fp32 MFMA score
    tiles against every item, history masked in the tile) -- users/s end to end and the rank kernel's own rate against the fp32 MFMA peak."""
    from idvs.morec_amd.data_utils.metrics import PackedEvalUsers, eval_ranks_packed, get_item_embeddings
    was_training = model.training
    out = {}
    try:
        model.eval()
        # first pass of the process = warm-up (an epoch loop evaluates every epoch: the steady state is "not the first pass"); it is timed too,
        # so that what a cold allocator costs on this box stays visible (BENCH_r05: 2.94 s for a pass whose kernels take 0.28 s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        get_item_embeddings(model, content, test_bs, args, True, dev)
        torch.cuda.synchronize()
        dt_first = time.perf_counter() - t0
        allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        del gemm_log[:]
        timing_on["v"] = True
        t0 = time.perf_counter()
        emb = get_item_embeddings(model, content, test_bs, args, True, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timing_on["v"] = False
        allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0
        fl = sum(g_[0] for g_ in gemm_log)
        ms = sum(g_[1].elapsed_time(g_[2]) for g_ in gemm_log)
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out["encode_all_items"] = {"items": int(content.shape[0]), "seconds": round(dt, 4), "items_per_s": round(content.shape[0] / dt, 1),
                                   "gemm_tflops": round(tf, 1), "gemm_frac_of_mfma_peak": round(tf / peak, 4), "gemm_ms": round(ms, 2),
                                   "chunk": test_bs, "first_pass_seconds": round(dt_first, 4), "device_allocations_in_pass": int(allocs),
                                   "note": "get_item_embeddings(use_modal=True): pinned-free H2D of each chunk's token rows + BERT forward on the real tokens "
                                           "+ fc/GELU head, eval mode, no grad; wall clock of the whole pass (second pass of the process; the chunks are encoded "
                                           "largest first, so a pass allocates nothing after its first chunk and nothing at all from the second pass on)"}
        # synthetic eval users: full-length sequences (S inputs + the target), history = the inputs (what run.py masks)
        rng = np.random.default_rng(777)
        seqs = rng.integers(1, item_num + 1, size=(n_users, S + 1))
        eval_seq = {u: seqs[u] for u in range(n_users)}
        hist = {u: seqs[u, :-1] for u in range(n_users)}
        users = list(range(n_users))
        packed = PackedEvalUsers(hist, eval_seq, users, S)
        eval_ranks_packed(model, packed, 0, test_bs, emb, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        packed = PackedEvalUsers(hist, eval_seq, users, S)      # the host-side packing is part of the pass
        t_pack = time.perf_counter() - t0
        for s0 in range(0, n_users, test_bs):
            eval_ranks_packed(model, packed, s0, min(n_users, s0 + test_bs), emb, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the rank kernel alone (launch sequence of the C-ABI call), HIP events
        prec = torch.randn(test_bs, D, device=dev)
        h_d = torch.from_numpy(packed.hist[:test_bs]).to(dev)
        t_d = torch.from_numpy(packed.target[:test_bs]).to(dev)
        ops.eval_rank(prec, emb, h_d, t_d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.eval_rank(prec, emb, h_d, t_d)
        e1.record()
        torch.cuda.synchronize()
        kms = e0.elapsed_time(e1) / 5
        kfl = 2.0 * test_bs * emb.shape[0] * D
        ktf = kfl / (kms * 1e-3) / 1e12
        out["rank_users"] = {"users": n_users, "items": int(emb.shape[0]), "seconds": round(dt, 4), "users_per_s": round(n_users / dt, 1),
                             "host_packing_s": round(t_pack, 4),
                             "rank_kernel": {"users_per_call": test_bs, "ms_per_call": round(kms, 3), "tflops": round(ktf, 1), "bound": "mfma (exact fp32)",
                                             "peak": MFMA_PEAK_TFLOPS["f32"], "frac": round(ktf / MFMA_PEAK_TFLOPS["f32"], 4),
                                             "scores_bytes_never_written": int(test_bs) * int(emb.shape[0]) * 4},
                             "note": "eval_model's inner loop: vectorised host packing of the users, item-vector gather, SASRec user encoder, morec_eval_rank "
                                     "(rank = 1 + #{unmasked items scoring above the target}; the [users, items] score matrix is never stored)"}
    except Exception as e:  # noqa: BLE001 -- a secondary line must never cost the headline
        out["error"] = f"{type(e).__name__}: {e}"
        timing_on["v"] = False
    if was_training:
        model.train()
    return out


def operand_activity_probe(ops, dev, dt, M=54919, N=3072, K=768, iters=20):
    """One GEMM launch of the step's FFN-up shape (plain epilogue) timed on random and on all-zero operands.  Same kernel, same cycles; a part
    that runs at its power limit clocks the quiet operands higher (MI355X_MICROARCH.md "DVFS give-back").  The ratio says how far the random-
    operand rate -- the one `roofline.achieved` is made of -- sits below what the same schedule does when power is not the limit."""
    out = {}
    c = torch.empty(M, N, device=dev, dtype=dt)
    for name in ("random", "zero"):
        if name == "random":
            a_, b_ = (torch.randn(M, K, device=dev) * 0.5).to(dt), (torch.randn(N, K, device=dev) * 0.5).to(dt)
        else:
            a_, b_ = torch.zeros(M, K, device=dev, dtype=dt), torch.zeros(N, K, device=dev, dtype=dt)
        for _ in range(5):
            ops.gemm_nt(a_, b_, out=c)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm_nt(a_, b_, out=c)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            best = us if best is None else min(best, us)
        out[name + "_operands"] = {"us": round(best, 1), "tflops": round(2.0 * M * N * K / best / 1e6, 1)}
    out["shape"] = {"M": M, "N": N, "K": K, "dtype": str(dt).replace("torch.", "")}
    out["zero_over_random"] = round(out["zero_operands"]["tflops"] / out["random_operands"]["tflops"], 3)
    out["note"] = ("same launch, same instruction stream: the gap is the clock the part sustains on each operand set, i.e. the random-operand rate is "
                   "power-limited by that factor (profiles/r06_gemm_power_probe.txt has both tile kernels and K = 3072)")
    return out


def scoring_pooled(ops, B, S, D, dev, peak, ranks=8, rank=3, iters=20, dt=torch.bfloat16):
    """Fused in-batch CE forward + backward at the pooled-negative size of an `ranks`-GPU step, bf16: call-level times (every launch of
    the C-ABI call: table prep, positive logits, tile kernel, combine / dl^T GEMMs) and the executed FLOP rate."""
    Nr, Nc = B * S, ranks * B * (S + 1)
    g = torch.Generator(device=dev).manual_seed(ranks)
    P = (torch.randn(Nr, D, device=dev, generator=g) * 0.3).to(dt)
    E = (torch.randn(Nc, D, device=dev, generator=g) * 0.3).to(dt)
    ids = torch.randint(1, 80000, (Nc,), device=dev, generator=g, dtype=torch.int32)
    row_ids = ids[rank * B * (S + 1):(rank + 1) * B * (S + 1)].contiguous()
    logpop = torch.randn(Nc, device=dev, generator=g) - 9.0
    col_valid = torch.ones(Nc, device=dev, dtype=torch.uint8)
    row_valid = torch.ones(Nr, device=dev, dtype=torch.uint8)
    desc = ops.ce_desc(B, S, D, Nc, rank * B * (S + 1), dt, dE_fp32=True)
    ws = ops.ce_workspace(desc, dev)
    fwd = lambda: ops.inbatch_ce_fwd(desc, P, E, row_ids, ids, logpop, col_valid, row_valid, ws)      # noqa: E731
    _, lse, _ = fwd()
    bdesc = ops.ce_desc(B, S, D, Nc, rank * B * (S + 1), dt, dE_fp32=True, ws_from_fwd=True)      # as engine.ce_backward calls it: the forward's tables reused
    bwd = lambda: ops.inbatch_ce_bwd(bdesc, P, E, row_ids, ids, logpop, col_valid, row_valid, lse, None, 1.0 / Nr, ws)   # noqa: E731
    bwd()
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * Nr * Nc * D
    alg_f = (Nr + Nc) * D * 2 + 13 * Nc + 8 * Nr
    return {"Nr": Nr, "Nc": Nc, "D": D, "fwd_us": round(res["fwd"], 1), "bwd_us": round(res["bwd"], 1),
            "fwd_tflops": round(fl / res["fwd"] / 1e6, 1), "bwd_tflops": round(3 * fl / res["bwd"] / 1e6, 1),
            "fwd_frac_of_mfma_peak": round(fl / res["fwd"] / 1e6 / peak, 4), "fwd_algorithmic_GBps": round(alg_f / res["fwd"] / 1e3, 1),
            "note": "call-level: ce8p prep + positive-logit + 256 x 256 tile kernel + combine (fwd); + dl^T, dE = dl^T P (one NT GEMM, fp32), "
                    "dP = dl E (transposing GEMM) (bwd); kernel-level numbers in profiles/"}


def cpu_baseline(a):
    """The CPU oracle (own restatement of the reference path, ``oracle/``; the reference's Python cannot travel to
    the GPU box) timed on the host cores over a bounded sample: a few user sequences, fwd + bwd + AdamW."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import morec_oracle as orc
    from idvs.morec_amd.model import BertShape
    from idvs.morec_amd.model.spec import model_param_shapes
    S, T, D = 20, 30, 512
    shape = BertShape.named(a.bert)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    Bc = a.cpu_batch
    rng = np.random.default_rng(1)
    content = synth_catalog(a.item_num, T, rng)
    n_max = 4                                # one untimed warm-up step + up to three timed ones
    ids = synth_batches(n_max, Bc, S, a.item_num, rng)
    counts = np.bincount(ids.reshape(-1), minlength=a.item_num + 1).astype(np.float64) + 1.0
    pop = counts / counts[1:].sum()
    pop[0] = 1.0
    shapes = model_param_shapes(max_seq_len=S, embedding_dim=D, n_blocks=2, item_num=a.item_num, use_modal=True, bert=shape)
    g = torch.Generator().manual_seed(0)
    p = {k: (torch.randn(*s, generator=g) * 0.02).requires_grad_(True) for k, s in shapes.items()}
    states = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
    # Bounded sample of the SAME workload (BERT-base, S = 20, T = 30, D = 512; Bc user sequences per step instead of 128): the first step
    # warms the allocator / thread pool and is not timed; then whole steps until ~20 s of timed CPU work (or three steps) are in.
    times = []
    t_begin = time.perf_counter()
    for it in range(n_max):
        t0 = time.perf_counter()
        items = torch.from_numpy(content[ids[it].reshape(-1)])
        loss = orc.model_forward(p, torch.from_numpy(ids[it]).view(-1), items, torch.ones(Bc, S), pop, max_seq_len=S,
                                 embedding_dim=D, n_heads=2, use_modal=True, bert_heads=shape.num_attention_heads)
        loss.backward()
        with torch.no_grad():
            for k, v in p.items():
                if v.grad is None or "pooler" in k:
                    continue
                lr = 5e-5 if "bert_model" in k else 1e-4
                orc.adamw_step(v, v.grad, states[k][0], states[k][1], it + 1, lr, 0.01)
                v.grad = None
        times.append(time.perf_counter() - t0)
        timed = sum(times[1:])
        if (len(times) >= 2 and (timed >= 20.0 or time.perf_counter() - t_begin + times[-1] > 0.8 * a.cpu_timeout)) or times[-1] > 60:
            break
    timed_steps = times[1:] if len(times) > 1 else times
    mean = sum(timed_steps) / len(timed_steps)
    return {"value": round(Bc / mean, 4), "unit": "user-seq/s", "cores": threads, "kind": "port",
            "sample": f"{Bc} user sequences per step x {len(timed_steps)} timed step(s) after one untimed warm-up step = {sum(timed_steps):.1f} s of CPU work "
                      f"(fwd + bwd + AdamW of the PyTorch-CPU fp32 oracle, BERT-{a.bert}, S = 20, T = 30, D = 512: the bench workload at batch {Bc} "
                      f"instead of 128); mean step {mean:.2f} s, best {min(timed_steps):.2f} s"}


if __name__ == "__main__":
    main()
