"""Evaluation with the reference's entry points (``T/data_utils/metrics.py``): ``get_item_embeddings`` and
``eval_model`` keep their signatures; the per-user Python loop + full argsort (``metrics.py:97-102,49-57``) is
replaced by the count-greater HIP kernel (``morec_eval_rank``), the score matrix is never materialised."""
from __future__ import annotations


import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from .dataset import SequentialDistributedSampler


def print_metrics(x, Log_file, v_or_t):
    Log_file.info(v_or_t + "_results   {}".format("\t".join(["{:0.5f}".format(i * 100) for i in x])))


def _module(model):
    return model.module if hasattr(model, "module") else model


def get_item_embeddings(model, item_content, test_batch_size, args, use_modal, local_rank):
    """``metrics.py:60-74``: encode every item (row 0 = padding item) in eval mode, no grad -> fp32 [item_num+1, D].
    Stays on the device (the reference round-trips through the CPU).  Runs under the model's fp32-GEMM mode (``compute_dtype``
    fp32x3: the encoders called on their own would otherwise fall back to the exact-fp32 MFMA)."""
    model.eval()
    with ops.fp32_gemm_mode(getattr(_module(model), "fp32_gemm", ops.FP32_GEMM)):
        return _get_item_embeddings(model, item_content, test_batch_size, args, use_modal, local_rank)


def _get_item_embeddings(model, item_content, test_batch_size, args, use_modal, local_rank):
    m = _module(model)
    vision = bool(use_modal and getattr(m, "vision", False))
    # vision (``get_itemLMDB_embeddings``, V/data_utils/metrics.py:63-76): ``item_content`` is the decoded image tensor
    # f32[item_num+1, 3, R, R] (the LMDB / PIL decode of V/data_utils/dataset.py is host-side I/O outside this library)
    if hasattr(item_content, "device_batch"):       # LmdbItemImages: records decoded per chunk, resized on the device
        outs = []
        with torch.no_grad():
            for s in range(0, len(item_content), test_batch_size):
                ids = np.arange(s, min(len(item_content), s + test_batch_size))
                outs.append(m.cv_encoder(item_content.device_batch(ids, local_rank).contiguous()))
        return torch.cat(outs, 0).float().detach()
    content = torch.as_tensor(np.asarray(item_content))
    if not vision:
        content = content.long()
    elif content.dtype != torch.uint8:      # uint8 HWC images are normalised on the device (morec_swin_patchify_u8)
        content = content.float()
    n = int(content.shape[0])
    starts = list(range(0, n, test_batch_size))
    if use_modal and not vision and content.dim() == 2 and len(starts) > 1:
        # The chunks are the reference's (consecutive slices of test_batch_size items); they are ENCODED largest first.  The text tower runs
        # on the real tokens only, so every chunk has its own row count, and a chunk bigger than every earlier one sends the caching
        # allocator to hipMalloc for each of its tensors (2.6 s of a 2.9 s pass on a box with slow page mapping: BENCH_r05).  With the
        # largest chunk first every later request fits a cached block: no allocation inside the pass after its first chunk, none at all from
        # the second pass on.  An item's vector does not depend on the chunk around it (row-wise operators; attention per title).
        weight = (content != 0).sum(1).numpy()  # rows are [ids | mask] per attribute, [PAD] = 0 in both: non-zero entries = 2 x real tokens
        tok = np.add.reduceat(weight, starts)
        starts = [starts[i] for i in np.argsort(-tok, kind="stable")]
    out = None
    with torch.no_grad():
        for s in starts:
            chunk = content[s:s + test_batch_size].to(local_rank)
            if vision:
                o = m.cv_encoder(chunk.contiguous())
            else:
                o = m.bert_encoder(chunk) if use_modal else m.id_embedding(chunk)
            if out is None:
                out = torch.empty((n,) + tuple(o.shape[1:]), device=o.device, dtype=torch.float32)
            out[s:s + o.shape[0]] = o
    return out.detach()


def metrics_from_ranks(ranks: torch.Tensor, topK: int = 10):
    """Hit@K and nDCG@K per user from 1-based ranks (``metrics_topK``, ``metrics.py:49-57``)."""
    hit = (ranks <= topK).float()
    ndcg = torch.where(ranks <= topK, 1.0 / torch.log2(ranks.float() + 1.0), torch.zeros_like(hit))
    return hit, ndcg


class PackedEvalUsers:
    """The evaluation inputs of a list of users as dense arrays, built ONCE per ``eval_model`` call with array operations (the reference
    walks the users one by one in Python, ``metrics.py:92-102``; so did the first version of this file, per batch):
    ``idx`` int64 [U, S] = the right-aligned input sequence ``seq[:-1]`` (0 = padding), ``lm`` float32 [U, S] its mask, ``target`` int32 [U]
    = ``seq[-1]``, ``hist`` int32 [U, Hmax] = the history items to mask, padded with -1."""

    def __init__(self, user_history, eval_seq, users, S):
        U = len(users)
        seq_len = np.fromiter((len(eval_seq[u]) for u in users), dtype=np.int64, count=U)
        if U and int(seq_len.min()) < 1:
            raise ValueError("an evaluation sequence needs at least its target item")
        flat = np.concatenate([np.asarray(eval_seq[u], dtype=np.int64) for u in users]) if U else np.zeros(0, np.int64)
        ends = np.cumsum(seq_len)
        starts = ends - seq_len
        self.target = flat[ends - 1].astype(np.int32) if U else np.zeros(0, np.int32)
        n_in = np.minimum(seq_len - 1, S)                        # inputs kept: the last S of seq[:-1] (sequences are <= S + 1 long by construction)
        rows = np.repeat(np.arange(U), n_in)
        k = np.arange(int(n_in.sum())) - np.repeat(np.cumsum(n_in) - n_in, n_in)          # 0 .. n_in-1 within each user
        src = np.repeat(ends - 1 - n_in, n_in) + k               # positions of those inputs in `flat`
        cols = np.repeat(S - n_in, n_in) + k                     # right-aligned
        self.idx = np.zeros((U, S), dtype=np.int64)
        self.lm = np.zeros((U, S), dtype=np.float32)
        self.idx[rows, cols] = flat[src]
        self.lm[rows, cols] = 1.0
        h_len = np.fromiter((len(user_history[u]) for u in users), dtype=np.int64, count=U)
        hmax = max(1, int(h_len.max()) if U else 1)
        self.hist = np.full((U, hmax), -1, dtype=np.int32)
        if U and int(h_len.sum()):
            hflat = np.concatenate([np.asarray(user_history[u], dtype=np.int64).reshape(-1) for u in users])
            hrows = np.repeat(np.arange(U), h_len)
            hcols = np.arange(int(h_len.sum())) - np.repeat(np.cumsum(h_len) - h_len, h_len)
            self.hist[hrows, hcols] = hflat.astype(np.int32)
        self.U, self.S = U, S

    def slice(self, a, b):
        hist = self.hist[a:b]
        used = max(1, int((hist >= 0).sum(1).max()) if b > a else 1)     # the chunk's own widest history: what the kernel's LDS table holds
        return self.idx[a:b], self.lm[a:b], np.ascontiguousarray(hist[:, :used]), self.target[a:b]


def eval_ranks_packed(model, packed: PackedEvalUsers, a, b, item_embeddings, local_rank):
    """1-based target ranks of users [a, b) of a ``PackedEvalUsers``: user states from the SASRec encoder, ranks by the HIP kernel."""
    m = _module(model)
    idx, lm, hist, target = packed.slice(a, b)
    dev = item_embeddings.device
    with torch.no_grad(), ops.fp32_gemm_mode(getattr(m, "fp32_gemm", ops.FP32_GEMM)):
        embs = item_embeddings[torch.from_numpy(idx).to(dev)]                      # [U, S, D] gather (plumbing)
        prec = m.user_encoder(embs, torch.from_numpy(lm).to(dev), local_rank)[:, -1].float().contiguous()
        return ops.eval_rank(prec, item_embeddings.contiguous(), torch.from_numpy(hist).to(dev),
                             torch.from_numpy(target).to(dev)).long()


def eval_ranks(model, user_history, eval_seq, item_embeddings, users, args, local_rank):
    """1-based target ranks for ``users`` (list of user ids)."""
    packed = PackedEvalUsers(user_history, eval_seq, users, args.max_seq_len)
    return eval_ranks_packed(model, packed, 0, packed.U, item_embeddings, local_rank)


def eval_model(model, user_history, eval_seq, item_embeddings, test_batch_size, args, item_num, Log_file, v_or_t, local_rank):
    """``metrics.py:77-107``: mean Hit@10 / nDCG@10 over all users (sharded over ranks like the reference's
    SequentialDistributedSampler, gathered with all_gather).  Returns Hit@10."""
    model.eval()
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    n_users = len(eval_seq)
    users_all = list(range(n_users))
    sampler = SequentialDistributedSampler(users_all, test_batch_size, rank=rank, num_replicas=world)
    mine = list(iter(sampler))
    item_embeddings = item_embeddings.to(local_rank)
    hits, ndcgs = [], []
    packed = PackedEvalUsers(user_history, eval_seq, mine, args.max_seq_len)      # one vectorised pass over this rank's users
    for s in range(0, len(mine), test_batch_size):
        ranks = eval_ranks_packed(model, packed, s, min(len(mine), s + test_batch_size), item_embeddings, local_rank)
        h, n = metrics_from_ranks(ranks)
        hits.append(h)
        ndcgs.append(n)
    hit, ndcg = torch.cat(hits), torch.cat(ndcgs)
    if world > 1:
        parts_h = [torch.empty_like(hit) for _ in range(world)]
        parts_n = [torch.empty_like(ndcg) for _ in range(world)]
        dist.all_gather(parts_h, hit)
        dist.all_gather(parts_n, ndcg)
        hit, ndcg = torch.cat(parts_h), torch.cat(parts_n)
    hit, ndcg = hit[:n_users], ndcg[:n_users]       # drop the sampler's padding (metrics.py:33-37 ``[:num_total_examples]``)
    mean_eval = [hit.mean().item(), ndcg.mean().item()]
    if Log_file is not None:
        Log_file.info(v_or_t + "_methods   {}".format("\t".join(["Hit10", "nDCG10"])))
        print_metrics(mean_eval, Log_file, v_or_t)
    return mean_eval[0]
